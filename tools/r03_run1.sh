#!/bin/bash
# round 3, GPU call 1: new tests (batched generate_single, graph reuse, RCCL world-size-1), the default bench line, config-1 kernel profile
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/r03a; mkdir -p $OUT; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_msa.py tests/test_gpu_config5_and_protocol.py tests/test_gpu_rccl_single.py tests/test_gpu_w16_kernel.py tests/test_gpu_strict_kernels.py -m gpu -x -q 2>&1 | tail -25 > $OUT/pytest_new.log
cat $OUT/pytest_new.log | tail -15
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 3000 $OUT/bench_default.json; echo; tail -5 $OUT/bench_default.err
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/profs && PGIBBS_SMALL_B=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o p -- python $ROOT/tools/bench_small.py > /tmp/profs.log 2>&1)
tail -3 /tmp/profs.log
cp $(find /tmp/profs -name "*kernel_stats.csv" | head -1) $OUT/cfg1_kernel_stats.csv 2>/dev/null
head -30 $OUT/cfg1_kernel_stats.csv | cut -c1-90,150-260
