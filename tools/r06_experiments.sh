#!/bin/bash
# The round-6 experiments as they were run on the GPU box (each section wrote the profiles/r06_* file named beside it).
#   bash tools/r06_experiments.sh ladder|fewchain|traffic|rowln|rowln_stagger|rowln_ablation|ab
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06x; mkdir -p $O
case "$1" in
  ladder)          # profiles/r06_ladder_bench.txt: every tile height forced at the residual / fc1 shapes of a 1/8 ... 1/1 shard
    python tools/ladder_bench.py | tee $O/ladder_bench.txt ;;
  fewchain)        # profiles/r06_few_chain_gemm_bench.txt: every tile kernel forced at 8 ... 32 chains
    python tools/few_chain_gemm_bench.py | tee $O/few_chain_gemm_bench.txt ;;
  traffic)         # profiles/r06_hbm_traffic_pmc.json, r06_gemm_pmc_counters.txt
    bash tools/pmc_traffic.sh r06 | tee $O/traffic.txt
    for k in gemm_bf16_w16_kernel gemm_bf16_pp_kernel attention_kernel layernorm_bf16_kernel; do echo "== $k"; bash tools/pmc_bench.sh $k r06$k; done 2>&1 | grep -E "^==|^pass" > $O/gemm_pmc_counters_raw.txt
    python tools/pmc_counters_report.py $O/gemm_pmc_counters_raw.txt "round 6" > $O/gemm_pmc_counters.txt ;;
  rowln)           # profiles/r06_rowln_bench.txt
    python tools/rowln_bench.py | tee $O/rowln_bench.txt ;;
  rowln_stagger)   # profiles/r06_rowln_stagger.txt: populations of first-round workgroups delayed against each other
    for pop in 1 2 3 4; do for us in 0 10 20 30 45; do
      [ $pop = 1 ] && [ $us != 0 ] && continue
      echo "== populations $pop, stagger ${us} us (0 = default period / populations)"
      PGIBBS_ROWLN_POP=$pop PGIBBS_ROWLN_STAGGER_US=$us python tools/rowln_bench.py 2>&1 | grep -v amdgpu | sed -n 2,3p
    done; done | tee $O/rowln_stagger.txt ;;
  rowln_ablation)  # profiles/r06_rowln_epilogue_ablation.txt
    for abl in 2 12 13 14; do
      echo "== epilogue ablation $abl (2 = 4 half-steps + whole epilogue, 12 = no LayerNorm arithmetic, 13 = no residual row loads, 14 = no global stores)"
      PGIBBS_ROWLN_POP=1 PGIBBS_ROWLN_BENCH_ABL=$abl python tools/rowln_bench.py 2>&1 | grep -v amdgpu | sed -n 2,2p
    done | tee $O/rowln_epilogue_ablation.txt ;;
  ab)              # profiles/r06_shard_proxy_ladder_ab.txt, r06_msa_cfg4_rowln_ab.txt: the two kernels in the engine, interleaved
    B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-host-entry"
    for i in 1 2; do
      PGIBBS_GEMM_LADDER=1 $B > $O/proxy_ladder_on_$i.json; $B > $O/proxy_ladder_off_$i.json
      PGIBBS_ROWLN=1 python bench_msa.py --config 4 --steps 3 --warmup 1 > $O/cfg4_rowln1_$i.json; python bench_msa.py --config 4 --steps 3 --warmup 1 > $O/cfg4_rowln0_$i.json
    done ;;
  *) echo "usage: $0 ladder|fewchain|traffic|rowln|rowln_stagger|rowln_ablation|ab" ;;
esac
