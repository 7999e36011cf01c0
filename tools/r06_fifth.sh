#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06e; mkdir -p $O
bash tools/pmc_traffic.sh r06 > $O/traffic.txt 2>&1; tail -32 $O/traffic.txt | cut -c1-220
for k in gemm_bf16_w16_kernel gemm_bf16_pp_kernel attention_kernel layernorm_bf16_kernel; do echo "== $k"; bash tools/pmc_bench.sh $k r06$k; done 2>&1 | grep -E "^==|^pass" > $O/gemm_pmc_counters_raw.txt
python tools/pmc_counters_report.py $O/gemm_pmc_counters_raw.txt "round 6" > $O/gemm_pmc_counters.txt; head -14 $O/gemm_pmc_counters.txt | cut -c1-220
