#!/bin/bash
# round 6 evidence set from one HEAD: full GPU test suite, smoke(), the default bench line, rocprofv3 kernel stats (configs 2, 4),
# the few-chain sweep, the N = 2 code path of bench.py on one GPU, the shard-proxy A/B of the attention split.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06ev; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python bench.py --gpus 2 --oversubscribe --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-roofline > $O/bench_n2_oversubscribed.json 2> $O/bench_n2.err; tail -c 300 $O/bench_n2_oversubscribed.json; echo
python bench.py --gpus 1 --force-dist --native-gather --steps 3 --warmup 1 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-roofline --no-shard-proxy > $O/bench_n1_rccl_native_gather.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1_rccl_native_gather.json; echo
bash tools/r03_prof.sh r06ev > $O/prof_cfg2.txt 2>&1; tail -14 $O/prof_cfg2.txt | cut -c1-170
bash tools/r03_prof_msa.sh r06ev 4 > $O/prof_msa4.txt 2>&1; tail -14 $O/prof_msa4.txt | cut -c1-170
python tools/batch_sweep.py > $O/batch_sweep.txt 2> $O/batch_sweep.err; cut -c1-200 $O/batch_sweep.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-host-entry"
for i in 1 2; do
  $B > $O/proxy_split_on_$i.json 2>> $O/err.txt
  PGIBBS_ATTN_SPLIT=0 $B > $O/proxy_split_off_$i.json 2>> $O/err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06ev/proxy_*.json")):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "failed", e); continue
    sp = d.get("shard_proxy", {}).get("shards", {})
    ts = d.get("time_split_ms_per_iter", {})
    print("%-28s full %.2f ms | %s | attn %.2f" % (f.split("/")[-1], d["ms_per_step"], "  ".join("%s: %.2f ms (%.3f)" % (k, v["ms_per_step"], v["efficiency_vs_linear"]) for k, v in sp.items()), ts.get("attention", 0)))
PY
