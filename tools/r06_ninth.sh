#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06i; mkdir -p $O
for abl in 2 12 13 14; do
  echo "== epilogue ablation $abl (2 = 4 half-steps + whole epilogue, 12 = no LayerNorm arithmetic, 13 = no residual row loads, 14 = no global stores)"
  PGIBBS_ROWLN_POP=1 PGIBBS_ROWLN_BENCH_ABL=$abl python tools/rowln_bench.py 2>&1 | grep -v amdgpu | sed -n 2,2p
done > $O/rowln_epilogue_ablation.txt 2>&1
cat $O/rowln_epilogue_ablation.txt
