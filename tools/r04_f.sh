#!/bin/bash
# round 4, run F: padded ragged MSA batches + the loglik recordings regenerated from the reference (list in one call)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04f; mkdir -p $O
python -m pytest tests/test_gpu_loglik.py -x -q > $O/test_loglik.txt 2>&1; tail -15 $O/test_loglik.txt
python -m pytest tests/test_gpu_msa.py tests/test_gpu_strict_kernels.py tests/test_gpu_kernels.py -x -q -k "msa or attention" > $O/test_regress.txt 2>&1; tail -3 $O/test_regress.txt
