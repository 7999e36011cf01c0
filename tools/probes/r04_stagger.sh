#!/bin/bash
# round 4, experiment B: de-phasing the CUs of an XCD in the 8-wave residual GEMM (start stagger), tails first / last
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04b; mkdir -p $O
export PGIBBS_GEMM_RESID=pp PGIBBS_BENCH_ITERS=150
for tl in 0 1; do for st in 0 10000 20000 30000 45000 60000; do
  echo "tail_last=$tl stagger=$st" >> $O/stagger.txt
  PGIBBS_GEMM_TAIL_LAST=$tl PGIBBS_PP_STAGGER=$st python tools/gemm_bench_r2.py 2 2>/dev/null | head -7 >> $O/stagger.txt
done; done
cat $O/stagger.txt
