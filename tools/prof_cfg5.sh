#!/bin/bash
# GPU box: per-kernel stats of one config-5 template (generate_single, depth 128 x L=512), row attention split by grid size
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o p -- python $GRAFT_REPO_ROOT/bench_msa.py --config 5 --templates 1 > /tmp/prof.log 2>&1
head -8 $(find /tmp/prof5 -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-260
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/prof5/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(f)):
    if "msa_row_attention" in r["Kernel_Name"]:
        k = (r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("Workgroup_Size_X", "?"))
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
for k, v in sorted(agg.items()):
    print("row attention grid", k, "n=%d avg %.1f us" % (v[1], v[0] / v[1] / 1e3))
PY
