#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06d; mkdir -p $O
python tools/few_chain_gemm_bench.py > $O/few_chain_gemm_bench.txt 2> $O/err.txt; cat $O/few_chain_gemm_bench.txt; tail -3 $O/err.txt
bash tools/pmc_traffic.sh r06 > $O/traffic.txt 2>&1; tail -32 $O/traffic.txt | cut -c1-200
