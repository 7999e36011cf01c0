#!/bin/bash
# GPU box: per-kernel stats of tools/bench_small.py (1, 8 and 32 chains of L=25, 20 Gibbs iterations each)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profs && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o p -- python $GRAFT_REPO_ROOT/tools/bench_small.py > /tmp/profs.log 2>&1
tail -3 /tmp/profs.log
head -16 $(find /tmp/profs -name "*kernel_stats.csv" | head -1) | cut -c1-75,120-200
