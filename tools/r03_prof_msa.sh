#!/bin/bash
# rocprofv3 kernel stats of bench_msa.py (arg: TAG, config)
TAG=$1; CFG=${2:-4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_$TAG && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $ROOT/bench_msa.py --config $CFG --steps 3 --warmup 1 > /tmp/prof_$TAG.log 2>&1)
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/msa_cfg${CFG}_kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/msa_cfg${CFG}_kernel_stats.csv')))
for r in rows[:16]:
    print('%-100s calls %5s avg %9.1f us  %5s%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
