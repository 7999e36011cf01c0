#!/bin/bash
# round 5 evidence set from one HEAD: full GPU test suite, the default bench line, a 50-iteration job, rocprofv3 kernel stats
# (configs 2, 1, 4 and strict), PMC fabric traffic (ESM-1b and MSA), PMC counters of the hot kernels, the shard regime, the N = 2
# code path of bench.py on one GPU (gloo, ranks share the device), smoke().  Everything the profiles/r05_* files are copied from.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r05ev; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-host-entry > $O/bench_50iters.json 2>> $O/bench_default.err
python bench.py --gpus 2 --oversubscribe --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-roofline > $O/bench_n2_oversubscribed.json 2> $O/bench_n2.err; tail -c 200 $O/bench_n2_oversubscribed.json; echo
bash tools/r03_prof.sh r05ev > $O/prof_cfg2.txt 2>&1; tail -16 $O/prof_cfg2.txt | cut -c1-170
bash tools/r03_prof_msa.sh r05ev 4 > $O/prof_msa4.txt 2>&1; tail -16 $O/prof_msa4.txt | cut -c1-170
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profc1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc1 -o p -- python $GRAFT_REPO_ROOT/tools/cfg1_probe.py > /tmp/profc1.log 2>&1; cp $(find /tmp/profc1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/cfg1_kernel_stats.csv )
bash tools/pmc_traffic.sh r05 > $O/traffic.txt 2>&1; tail -14 $O/traffic.txt | cut -c1-170
python tools/shard_regime.py --engine > $O/shard_regime.txt 2>&1; tail -8 $O/shard_regime.txt
bash tools/pmc_traffic_msa.sh r05 > $O/traffic_msa.txt 2>&1; tail -8 $O/traffic_msa.txt | cut -c1-150
for k in gemm_bf16_w16_kernel gemm_bf16_pp_kernel attention_kernel; do echo "== $k"; bash tools/pmc_bench.sh $k r05$k; done 2>&1 | grep -E "^==|^pass" > $O/gemm_pmc_counters.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_strict && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_strict -o p -- python $GRAFT_REPO_ROOT/bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-msa --no-fp16 --no-roofline --no-host-entry > /tmp/prof_strict.log 2>&1; cp $(find /tmp/prof_strict -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/strict_cfg2_kernel_stats.csv )
