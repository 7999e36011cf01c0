#!/bin/bash
# round 4: device-wide barrier probe + the persistent single-chain trunk (bit-identity test, config-1 timing, kernel stats)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04i; mkdir -p $O
timeout 120 ./build/grid_barrier_bench > $O/barrier.txt 2>&1; cat $O/barrier.txt
( timeout 900 python -m pytest tests/test_gpu_chain_trunk.py -x -q ) > $O/test.txt 2>&1; tail -15 $O/test.txt
for v in 1 0 1 0; do echo "PGIBBS_CHAIN_TRUNK=$v"; PGIBBS_CHAIN_TRUNK=$v timeout 300 python tools/cfg1_probe.py 2>&1 | grep "stream=own" | tail -2; done > $O/cfg1.txt 2>&1; cat $O/cfg1.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profc1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc1 -o p -- python $GRAFT_REPO_ROOT/tools/cfg1_probe.py > /tmp/profc1.log 2>&1; cp $(find /tmp/profc1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/cfg1_kernel_stats.csv )
head -8 $O/cfg1_kernel_stats.csv | cut -c1-200
