#!/bin/bash
# round 6, second GPU call: the new kernels' tests, shard proxy A/B (ladder + attention split on / off), sweep with per-projection times
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06b; mkdir -p $O
( time python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16_mode.py tests/test_gpu_fullsize.py -x -q ) > $O/pytest_subset.txt 2>&1; tail -15 $O/pytest_subset.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-host-entry"
for i in 1 2; do
  $B > $O/proxy_new_$i.json 2> $O/err.txt
  PGIBBS_GEMM_LADDER=0 PGIBBS_ATTN_SPLIT=0 $B > $O/proxy_old_$i.json 2>> $O/err.txt
done
PGIBBS_GEMM_LADDER=0 $B > $O/proxy_noladder.json 2>> $O/err.txt
PGIBBS_ATTN_SPLIT=0 $B > $O/proxy_nosplit.json 2>> $O/err.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06b/proxy_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "failed", e); continue
    sp = d.get("shard_proxy", {}).get("shards", {})
    ts = d.get("time_split_ms_per_iter", {})
    print("%-28s full %.2f ms | %s | attn %.2f ln %.2f gemm %.2f" % (f.split("/")[-1], d["ms_per_step"],
          "  ".join("%s: %.2f ms (%.3f)" % (k, v["ms_per_step"], v["efficiency_vs_linear"]) for k, v in sp.items()),
          ts.get("attention", 0), ts.get("layernorm", 0), ts.get("gemm", 0)))
PY
python tools/batch_sweep.py --chains 8,16,32,64,128 --lengths 256 > $O/sweep_new.txt 2>> $O/err.txt
PGIBBS_GEMM_LADDER=0 PGIBBS_ATTN_SPLIT=0 python tools/batch_sweep.py --chains 8,16,32,64,128 --lengths 256 > $O/sweep_old.txt 2>> $O/err.txt
cut -c1-250 $O/sweep_new.txt; cut -c1-250 $O/sweep_old.txt
tail -5 $O/err.txt
