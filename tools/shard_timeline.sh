#!/bin/bash
# Where a 1/N shard's iteration goes: rocprofv3 kernel TRACE (start / end stamps of every dispatch) of bench.py --chains C, then
# busy time vs gaps between consecutive dispatches, per kernel.  args: TAG CHAINS [extra bench args]
TAG=$1; CH=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/tl_$TAG && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o p -- python $ROOT/bench.py --chains $CH --steps 6 --warmup 2 --no-cpu-baseline --no-strict --no-fp16 --no-msa --no-roofline --no-host-entry "$@" > /tmp/tl_$TAG.log 2>&1)
tail -1 /tmp/tl_$TAG.log | cut -c1-200
python3 - "$(find /tmp/tl_$TAG -name '*kernel_trace.csv' | head -1)" "$OUT/shard_timeline_${CH}chains.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows))
# steady state: the last 60 % of the dispatches (warm-up, weight upload and conversion kernels come first)
ev = ev[int(len(ev) * 0.4):]
span = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
per = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, n) in enumerate(ev):
    per[n][0] += 1; per[n][1] += e - s
    if i: per[n][2] += max(0, s - ev[i - 1][1])
out = ["dispatches %d, span %.3f ms, kernels busy %.3f ms (%.1f %%), gaps %.3f ms" % (len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6),
       "%-64s %6s %10s %10s %10s" % ("kernel", "calls", "avg us", "gap before", "share %")]
for n, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    out.append("%-64s %6d %10.1f %10.1f %10.1f" % (n[:64], c, d / c / 1e3, g / c / 1e3, 100.0 * (d + g) / span))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
PY
