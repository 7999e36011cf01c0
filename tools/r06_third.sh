#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06c; mkdir -p $O
python tools/ladder_bench.py > $O/ladder_bench.txt 2> $O/err.txt; cat $O/ladder_bench.txt; tail -3 $O/err.txt
