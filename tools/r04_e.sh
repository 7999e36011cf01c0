#!/bin/bash
# round 4, run E: ESM-1 architecture tests + regression of the attention kernels (bias-key parameter added)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04e; mkdir -p $O
python -m pytest tests/test_gpu_esm1.py -x -q -s > $O/test_esm1.txt 2>&1; tail -25 $O/test_esm1.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_strict_kernels.py tests/test_gpu_fp16_mode.py tests/test_gpu_engine.py tests/test_gpu_loglik.py -x -q > $O/test_regress.txt 2>&1; tail -3 $O/test_regress.txt
