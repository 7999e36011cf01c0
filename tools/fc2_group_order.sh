#!/bin/bash
# fc2 (M = 66048, N = 1280, K = 5120, fp32 residual epilogue, gemm_bf16_pp_kernel) reads 2.7x its geometric
# floor through the fabric.  Time and FETCH_SIZE for the m-panels-per-group settings the kernel is instantiated with
# (PGIBBS_GEMM_GM = 1 / 2 / 4; 2 is the default for K >= 4096).  FETCH_SIZE units -> bytes with the LayerNorm calibration of
# profiles/r04_hbm_traffic_pmc.json (2498 B per unit).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fc2_once.py <<PY
import ctypes, os, sys
sys.path.insert(0, "$ROOT")
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
it = int(sys.argv[1])
for name, M, N, K in (("fc2", 66048, 1280, 5120), ("out", 66048, 1280, 1280)):
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, 2, 20, it, ctypes.byref(ms)))
    print("GM=%s %s %.1f us %.0f TF" % (os.environ.get("PGIBBS_GEMM_GM", "default"), name, 1e3 * ms.value, 2.0 * M * N * K / ms.value / 1e9))
PY
for rep in 1 2; do for GM in 2 1 4; do PGIBBS_GEMM_GM=$GM python /tmp/fc2_once.py 200 2>&1 | grep GM=; done; done
for GM in 2 1 4; do
  rm -rf /tmp/fc2pmc_$GM
  PGIBBS_GEMM_GM=$GM rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fc2pmc_$GM -o p -- python /tmp/fc2_once.py 3 > /tmp/fc2pmc.log 2>&1
  python3 - $GM <<'PY'
import csv, glob, sys, collections
gm = sys.argv[1]
f = glob.glob("/tmp/fc2pmc_%s/**/*counter_collection.csv" % gm, recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f[0])):
    if row["Counter_Name"] == "FETCH_SIZE" and "gemm_bf16_pp" in row["Kernel_Name"]:
        k = row["Kernel_Name"].split("(")[0]
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
for k, v in agg.items():
    print("GM=%s %-60s launches %d FETCH_SIZE %.0f units = %.0f MB read per launch" % (gm, k[:60], v[1], v[0] / v[1], v[0] / v[1] * 2498.0 / 1e6))
PY
done
