#!/bin/bash
# usage: tools/pmc_gemm.sh VARIANT TAG   (on the GPU box; writes gpurun_out/pmc_TAG_passN.csv summaries)
V=$1; TAG=$2; ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
P2="SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"
P3="TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY GRBM_GUI_ACTIVE GRBM_TA_BUSY"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- python $ROOT/tools/gemm_prof.py $V > /tmp/pmc_run.log 2>&1
  python - "$i" "$TAG" <<'PY'
import csv, glob, sys, collections
i, tag = sys.argv[1], sys.argv[2]
f = glob.glob("/tmp/pmc_%s_%s/**/*counter_collection.csv" % (tag, i), recursive=True)
if not f:
    print("no counter csv; files:", glob.glob("/tmp/pmc_%s_%s/**/*" % (tag, i), recursive=True)); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f[0])):
    if "gemm" not in row["Kernel_Name"]:
        continue
    k = row["Counter_Name"]
    agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
print("pass", i, {k: v[0] / max(v[1], 1) for k, v in agg.items()})
PY
done
tail -1 /tmp/pmc_run.log
