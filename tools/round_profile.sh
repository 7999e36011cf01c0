#!/bin/bash
# GPU box: the evidence set of one round -> gpurun_out/<TAG>/  (copy what should be judged into profiles/):
#   pytest -m gpu log (with the parity figures the tests print), default bench JSON, rocprofv3 --kernel-trace --stats of the
#   same bench command, PMC fabric traffic per kernel (separate --pmc passes, tools/pmc_traffic.sh), config-4/5 MSA bench.
TAG=${1:-r03}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "it/s\]\|^job[0-9]\|^first\|^second\|^m1\|^Expected 2\|^bad line" > $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json; echo
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-shard-proxy --no-strict --no-msa > /tmp/prof_$TAG.log 2>&1)
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/esm1b_cfg2_kernel_stats.csv 2>/dev/null
head -12 $OUT/esm1b_cfg2_kernel_stats.csv | cut -c1-70,150-230
bash tools/pmc_traffic.sh $TAG > $OUT/pmc_traffic.log 2>&1; cp gpurun_out/traffic_$TAG.json $OUT/hbm_traffic_pmc.json 2>/dev/null; tail -4 $OUT/pmc_traffic.log
python bench_msa.py > $OUT/msa_bench.jsonl 2> $OUT/msa_bench.err; cut -c1-400 $OUT/msa_bench.jsonl
python bench_msa.py --precision fp32 --steps 2 > $OUT/msa_bench_strict.jsonl 2>> $OUT/msa_bench.err; cut -c1-400 $OUT/msa_bench_strict.jsonl
