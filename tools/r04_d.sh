#!/bin/bash
# round 4, run D: strict attention with one K/V staging per (sequence, head) (NQB = 1 old / 3 / 5); fp16 vs bf16 interleaved
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_gpu_strict_kernels.py -x -q > $O/test_strict.txt 2>&1; tail -3 $O/test_strict.txt
for nqb in 1 3 5 1 5; do
  PGIBBS_ATTN_F32_NQB=$nqb python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-msa > $O/strict_nqb$nqb.json 2> $O/strict.err
  python - <<PY
import json
d=json.loads(open("$O/strict_nqb$nqb.json").read().strip().splitlines()[-1])
print("strict nqb=$nqb", round(d["ms_per_step"],2), d.get("time_split_ms_per_iter"))
PY
done
for p in bf16 fp16 bf16 fp16; do
  python bench.py --precision $p --steps 10 --warmup 3 --no-cpu-baseline --no-msa --no-strict --no-fp16 > $O/ab_$p.json 2> $O/ab.err
  python - <<PY
import json
d=json.loads(open("$O/ab_$p.json").read().strip().splitlines()[-1])
print("$p", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["time_split_ms_per_iter"].items() if not isinstance(v, dict)}, {k: round(v["avg_launch_us"],1) for k,v in d.get("roofline",{}).get("per_kernel",{}).items()})
PY
done
python -m pytest tests/test_gpu_fullsize_logits.py -x -q -s -k "fp32" 2>&1 | grep -E "^\[|passed|failed" 
