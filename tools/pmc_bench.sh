#!/bin/bash
# usage: tools/pmc_bench.sh KERNEL_SUBSTRING TAG [bench args]   (GPU box): per-kernel PMC averages from a short bench run
KSUB=$1; TAG=$2; shift 2; ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P3="TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE GRBM_TA_BUSY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmcb_${TAG}_$i -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-shard-proxy --no-roofline --no-strict --no-msa --no-fp16 --no-host-entry --layers 3 "$@" > /tmp/pmcb_run.log 2>&1
  python - "$i" "$TAG" "$KSUB" <<'PY'
import csv, glob, sys, collections
i, tag, ksub = sys.argv[1:4]
f = glob.glob("/tmp/pmcb_%s_%s/**/*counter_collection.csv" % (tag, i), recursive=True)
if not f:
    print("no counter csv", glob.glob("/tmp/pmcb_%s_%s/**/*" % (tag, i), recursive=True)); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f[0])):
    if ksub not in row["Kernel_Name"]:
        continue
    k = row["Counter_Name"]
    agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
print("pass", i, {k: round(v[0] / max(v[1], 1)) for k, v in agg.items()}, "n=", max([v[1] for v in agg.values()] or [0]))
PY
done
