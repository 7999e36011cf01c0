#!/bin/bash
# usage: tools/pmc_msa.sh KERNEL_SUBSTRING   (GPU box): per-kernel PMC averages from one config-4 MSA iteration
KSUB=$1; ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P3="TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE GRBM_TA_BUSY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcm_$i -o p -- python $ROOT/bench_msa.py --config 4 --steps 1 --warmup 0 > /tmp/pmcm_run.log 2>&1
  python - "$i" "$KSUB" <<'PY'
import csv, glob, sys, collections
i, ksub = sys.argv[1:3]
f = glob.glob("/tmp/pmcm_%s/**/*counter_collection.csv" % i, recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f[0])):
    if ksub not in row["Kernel_Name"]:
        continue
    agg[row["Counter_Name"]][0] += float(row["Counter_Value"]); agg[row["Counter_Name"]][1] += 1
print("pass", i, {k: round(v[0] / max(v[1], 1)) for k, v in agg.items()}, "n=", max([v[1] for v in agg.values()] or [0]))
PY
done
