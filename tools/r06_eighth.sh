#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06h; mkdir -p $O
for pop in 1 2 3 4; do for us in 0 10 20 30 45; do
  [ $pop = 1 ] && [ $us != 0 ] && continue
  echo "== populations $pop, stagger ${us} us (0 = default period / populations)"
  PGIBBS_ROWLN_POP=$pop PGIBBS_ROWLN_STAGGER_US=$us python tools/rowln_bench.py 2>&1 | grep -v amdgpu | sed -n 2,3p
done; done > $O/rowln_stagger.txt 2>&1
cat $O/rowln_stagger.txt
