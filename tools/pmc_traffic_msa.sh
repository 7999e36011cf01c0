#!/bin/bash
# Fabric (L2 <-> Infinity Cache / HBM) traffic per kernel of one ESM-MSA-1b config-4 iteration, separate --pmc passes
#   tools/pmc_traffic_msa.sh TAG -> gpurun_out/traffic_msa_TAG.json
# Round 6: bytes straight from the request counters, as tools/pmc_traffic.sh (see there for the checks): reads = FETCH_SIZE KiB x 2
# (every read request is a 128-B request tallied at 64 B), writes >= WRITE_SIZE KiB (every write request tallied at 64 B: a lower
# bound).  The LayerNorm-derived scale factors of rounds 4-5 made every figure ~22 % too high.
TAG=${1:-r04}; ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/trafm_${TAG}_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/trafm_${TAG}_$C -o p -- python $ROOT/bench_msa.py --config 4 --steps 1 --warmup 0 > /tmp/trafm_run.log 2>&1
done
mkdir -p $ROOT/gpurun_out
python - "$TAG" "$ROOT" <<'PY'
import csv, glob, json, sys, collections
tag, root = sys.argv[1:3]
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/trafm_%s_%s/**/*counter_collection.csv" % (tag, c), recursive=True)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] != c:
            continue
        k = row["Kernel_Name"].split("(")[0]
        agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    raw[c] = {k: (v[0] / v[1], v[1]) for k, v in agg.items()}
M, d = 64 * 32 * 257, 768
ln = [k for k in raw["FETCH_SIZE"] if k.endswith("layernorm_bf16_kernel")][0]
out = {"method": "reads = FETCH_SIZE KiB x 2048 bytes (128-B requests tallied at 64 B), writes >= WRITE_SIZE KiB x 1024 (lower bound: 128-B "
                 "write requests tallied at 64 B); no scale factors (tools/pmc_traffic.sh documents the checks)",
       "layernorm_check": {"kernel": ln, "read_MB": raw["FETCH_SIZE"][ln][0] * 2048 / 1e6, "algorithmic_read_MB": M * d * 4 / 1e6,
                           "write_MB_lower_bound": raw["WRITE_SIZE"][ln][0] * 1024 / 1e6, "algorithmic_write_MB": M * d * 2 / 1e6},
       "kernels": {}}
for k in raw["FETCH_SIZE"]:
    fr, n = raw["FETCH_SIZE"][k]
    wr = raw["WRITE_SIZE"].get(k, (0.0, 0))[0]
    out["kernels"][k] = {"launches": n, "read_MB_per_launch": fr * 2048 / 1e6, "write_MB_per_launch_lower_bound": wr * 1024 / 1e6,
                         "write_MB_per_launch": wr * 1024 / 1e6}
json.dump(out, open("%s/gpurun_out/traffic_msa_%s.json" % (root, tag), "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["read_MB_per_launch"] * kv[1]["launches"]):
    print("%-70s n=%3d read %8.1f MB write >= %8.1f MB" % (k[:70], v["launches"], v["read_MB_per_launch"], v["write_MB_per_launch"]))
print("LayerNorm check:", out["layernorm_check"])
PY
