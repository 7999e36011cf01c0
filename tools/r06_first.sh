#!/bin/bash
# round 6, first GPU call: counter list, baseline bench line with the new fields, few-chain sweep, PMC traffic (old calibration)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06a; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>&1 | grep -E "TCC_EA|WRITE_SIZE|FETCH_SIZE|TCC_REQ|TCC_WRITE|TCC_READ" | head -150 ) > $O/counters.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json; echo
python tools/batch_sweep.py > $O/batch_sweep.txt 2> $O/batch_sweep.err; cat $O/batch_sweep.txt | cut -c1-200
bash tools/pmc_traffic.sh r06a > $O/traffic.txt 2>&1; tail -14 $O/traffic.txt | cut -c1-170
