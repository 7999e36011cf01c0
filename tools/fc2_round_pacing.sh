#!/bin/bash
# Round pacing of the deep-K residual GEMM (gemm_bf16_pp_kernel, PGIBBS_GEMM_RSYNC): time + FETCH_SIZE with and without, full-size
# fc2 (ESM-1b: M = 66048, N = 1280, K = 5120; ESM-MSA-1b: M = 526336, N = 768, K = 3072 with PGIBBS_GEMM_RSYNC_K=3072) through the
# engine's own dispatch (pg_dbg_gemm_bench variant 2).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/rs_once.py <<PY
import ctypes, os, sys
sys.path.insert(0, "$ROOT")
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
it = int(sys.argv[1])
shapes = (("fc2", 66048, 1280, 5120), ("fc2 64 chains", 16640, 1280, 5120), ("out", 66048, 1280, 1280)) if len(sys.argv) < 3 else (("msa fc2", 526336, 768, 3072),)
for name, M, N, K in shapes:
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, 2, 2, it, ctypes.byref(ms)))
    print("RSYNC=%s K>=%s %-14s M=%d N=%d K=%d: %.1f us %.0f TF" % (os.environ.get("PGIBBS_GEMM_RSYNC", "1"), os.environ.get("PGIBBS_GEMM_RSYNC_K", "4096"), name, M, N, K, 1e3 * ms.value, 2.0 * M * N * K / ms.value / 1e9))
PY
for rep in 1 2 3; do for RS in 0 1; do PGIBBS_GEMM_RSYNC=$RS python /tmp/rs_once.py 200 2>&1 | grep RSYNC=; done; done
for rep in 1 2; do for RS in 0 1; do PGIBBS_GEMM_RSYNC=$RS PGIBBS_GEMM_RSYNC_K=3072 python /tmp/rs_once.py 60 msa 2>&1 | grep RSYNC=; done; done
for RS in 0 1; do
  rm -rf /tmp/rspmc_$RS
  PGIBBS_GEMM_RSYNC=$RS rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rspmc_$RS -o p -- python /tmp/rs_once.py 3 > /tmp/rspmc.log 2>&1
  python3 - $RS <<'PY'
import csv, glob, sys, collections
rs = sys.argv[1]
f = glob.glob("/tmp/rspmc_%s/**/*counter_collection.csv" % rs, recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16_pp" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"].split("(")[0], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for (k, g), v in agg.items():
    print("RSYNC=%s %-52s grid %-9s launches %2d FETCH_SIZE %.0f units = %.0f MB read per launch" % (rs, k[:52], g, len(v), sum(v) / len(v), sum(v) / len(v) * 2498.0 / 1e6))
PY
done
