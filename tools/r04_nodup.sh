#!/bin/bash
# strict mode: producers skip the duplicate hi block of the split operand rows when the consumer is the fused kernel
# (PGIBBS_SPLIT3_NODUP=1, default) against writing all three blocks (=0).  Bit-identity test, then config 2 strict A/B.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04nd; mkdir -p $O
python -m pytest tests/test_gpu_strict_kernels.py tests/test_gpu_fullsize_logits.py tests/test_gpu_msa.py tests/test_gpu_engine.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
for nd in 1 0 1 0; do
  PGIBBS_SPLIT3_NODUP=$nd python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-msa --no-fp16 > $O/strict_nd$nd.json 2> $O/strict.err
  python - <<PY
import json
d=json.loads(open("$O/strict_nd$nd.json").read().strip().splitlines()[-1])
print("strict nodup=$nd", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["time_split_ms_per_iter"].items() if not isinstance(v, dict)})
PY
done 2>&1 | tee $O/ab.txt
