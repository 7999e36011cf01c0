#!/bin/bash
# rocprofv3 kernel stats of the config-2 bench (args: TAG, extra env as VAR=VAL ...)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard-proxy --no-strict --no-fp16 --no-msa --no-roofline --no-host-entry > /tmp/prof_$TAG.log 2>&1)
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/esm1b_cfg2_kernel_stats.csv 2>/dev/null
tail -2 /tmp/prof_$TAG.log | cut -c1-300
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/esm1b_cfg2_kernel_stats.csv')))
for r in rows[:14]:
    print('%-110s calls %5s avg %9.1f us  %5s%%' % (r['Name'][:110], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
