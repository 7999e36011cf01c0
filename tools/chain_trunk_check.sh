#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04m; mkdir -p $O
for g in 256 128; do timeout 60 ./build/chain_trunk_phases $g; done > $O/phases.txt 2>&1; cat $O/phases.txt
( timeout 1200 python -m pytest tests/test_gpu_chain_trunk.py -x -q ) > $O/test.txt 2>&1; tail -12 $O/test.txt
for v in 1 0 1; do echo "PGIBBS_CHAIN_TRUNK=$v"; PGIBBS_CHAIN_TRUNK=$v timeout 300 python tools/cfg1_probe.py 2>&1 | grep "stream=own" | tail -2; done > $O/cfg1.txt 2>&1; cat $O/cfg1.txt
