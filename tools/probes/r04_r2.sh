#!/bin/bash
# round 4, experiment A: the two-resident residual GEMM -- bit-identity, kernel times, whole-iteration A/B
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04a; mkdir -p $O
python tools/r2_check.py > $O/check.txt 2>&1
tail -3 $O/check.txt
python tools/gemm_bench_r2.py > $O/gemm.txt 2>&1
cat $O/gemm.txt
for st in 8000 20000 40000; do
  echo "stagger $st" >> $O/gemm_stagger.txt
  PGIBBS_R2_STAGGER=$st PGIBBS_BENCH_ITERS=100 python tools/gemm_bench_r2.py 50 2>&1 | head -4 >> $O/gemm_stagger.txt
done
cat $O/gemm_stagger.txt
for mode in pp r2 pp r2; do
  PGIBBS_GEMM_RESID=$mode python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-msa > $O/bench_$mode.json 2> $O/bench_$mode.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"), d.get("time_split_ms"))
PY
done
for mode in pp r2; do
  PGIBBS_GEMM_RESID=$mode python bench_msa.py --config 4 --steps 5 > $O/msa4_$mode.json 2> $O/msa4_$mode.err; tail -1 $O/msa4_$mode.json
done
