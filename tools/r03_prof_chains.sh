#!/bin/bash
# rocprofv3 kernel stats of bench.py --chains N (one GPU's share of a strong-scaling job); args: TAG CHAINS
TAG=$1; CH=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $ROOT/bench.py --chains $CH --steps 10 --warmup 2 --no-cpu-baseline --no-strict --no-msa --no-roofline > /tmp/prof_$TAG.log 2>&1)
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/esm1b_${CH}chains_kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/esm1b_${CH}chains_kernel_stats.csv')))
for r in rows[:10]:
    print('%-60s calls %5s avg %8.1f us  %5s%%' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
