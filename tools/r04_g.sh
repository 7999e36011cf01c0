#!/bin/bash
# round 4, run G: config 1 with the 16-wave deep-K fc2 (no split-K + reduce) vs the K-split form; kernel table of the new form
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > $O/test_gemm.txt 2>&1; tail -2 $O/test_gemm.txt
python -m pytest tests/test_gpu_fullsize_logits.py -x -q -k "config1" > $O/test_cfg1.txt 2>&1; tail -2 $O/test_cfg1.txt
for d in 0 1 0 1; do echo "PGIBBS_SKINNY_DEEP16=$d"; PGIBBS_SKINNY_DEEP16=$d python tools/cfg1_probe.py 2>&1 | tail -3; done
export TMPDIR=/tmp; cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg1 -- python $GRAFT_REPO_ROOT/tools/cfg1_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/cfg1_kernel_stats.csv; head -12 $O/cfg1_kernel_stats.csv | cut -c1-160
