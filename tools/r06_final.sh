#!/bin/bash
# final-HEAD check: the whole GPU suite, smoke(), the default bench line (timed), the shard regime tool
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06fin; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; grep -v amdgpu $O/pytest_gpu.txt | tail -6
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; cat $O/bench_time.txt; tail -c 200 $O/bench_default.json; echo
bash tools/pmc_traffic_msa.sh r06 > $O/traffic_msa.txt 2>&1; tail -12 $O/traffic_msa.txt | cut -c1-200
