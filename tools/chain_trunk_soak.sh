#!/bin/bash
# soak of the persistent trunk (bit identity over many calls at full depth) and two processes sharing the GPU
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04r; mkdir -p $O
for ct in 1 0; do PGIBBS_CHAIN_TRUNK=$ct timeout 600 python tools/probes/chain_trunk_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -3; done > $O/soak.txt 2>&1; cat $O/soak.txt
echo "two processes at once, persistent trunk on:" > $O/two.txt
(PGIBBS_CHAIN_TRUNK=1 timeout 600 python tools/probes/chain_trunk_soak.py 150 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/two.txt) &
(PGIBBS_CHAIN_TRUNK=1 timeout 600 python tools/probes/chain_trunk_soak.py 150 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/two.txt) &
wait
echo "two processes at once, per-layer launches:" >> $O/two.txt
(PGIBBS_CHAIN_TRUNK=0 timeout 600 python tools/probes/chain_trunk_soak.py 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/two.txt) &
(PGIBBS_CHAIN_TRUNK=0 timeout 600 python tools/probes/chain_trunk_soak.py 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/two.txt) &
wait
cat $O/two.txt
