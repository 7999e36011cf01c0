#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04k; mkdir -p $O
for e in "PGIBBS_CHAIN_TRUNK=0" "PGIBBS_CHAIN_TRUNK=1" "PGIBBS_CHAIN_TRUNK=1 PGIBBS_GRAPH=0" "PGIBBS_CHAIN_TRUNK=1 PRE=1" "PGIBBS_CHAIN_TRUNK=0 PRE=1" "PGIBBS_CHAIN_TRUNK=0 PGIBBS_GRAPH=0"; do echo "== $e"; env $e timeout 300 python tools/probes/chain_trunk_gen.py 2>&1 | grep -v amdgpu.ids | tail -4; done > $O/gen.txt 2>&1; cat $O/gen.txt
for g in 64 128 256; do echo "GRID=$g"; PGIBBS_CHAIN_TRUNK=1 PGIBBS_CHAIN_TRUNK_GRID=$g timeout 300 python tools/cfg1_probe.py 2>&1 | grep "stream=own" | tail -1; done > $O/grid.txt 2>&1; cat $O/grid.txt
