#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04w; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profc1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc1 -o p -- python $GRAFT_REPO_ROOT/tools/cfg1_probe.py > /tmp/profc1.log 2>&1; cp $(find /tmp/profc1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/cfg1_kernel_stats.csv )
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/cfg1_kernel_stats.csv')))
for r in rows[:22]:
    print('%-90s calls %5s avg %9.1f us  %5s%%' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
