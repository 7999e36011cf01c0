#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06g; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_msa.py -x -q -k "full_row" ) > $O/pytest_rowln.txt 2>&1; tail -3 $O/pytest_rowln.txt
for i in 1 2; do
  for sw in 1 0; do
    PGIBBS_ROWLN=$sw timeout 600 python bench_msa.py --config 4 --steps 3 --warmup 1 > $O/cfg4_rowln${sw}_$i.json 2>> $O/err.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06g/cfg*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "failed", e); continue
    ts = d.get("time_split_ms_per_iter") or d.get("time_split_ms_per_forward")
    print("%-28s %8.2f ms  %s" % (f.split("/")[-1], d.get("ms_per_step") or d.get("ms_per_template_forward"), {k: round(v, 2) for k, v in ts.items()}))
PY
tail -5 $O/err.txt
