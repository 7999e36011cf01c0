#!/bin/bash
# round 3, GPU call: LayerNorm fold + tail tiles: kernel/engine parity, shard invariance, bench
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/r03b; mkdir -p $OUT; cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_ln_fused.py tests/test_gpu_kernels.py tests/test_gpu_w16_kernel.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -30 > $OUT/pytest_fold.log
tail -22 $OUT/pytest_fold.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_msa.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest_full.log
tail -8 $OUT/pytest_full.log
timeout 900 python bench.py --no-cpu-baseline --no-strict --no-msa > $OUT/bench_fold.json 2> $OUT/bench_fold.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b/bench_fold.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_split_ms_per_iter'])
PY
PGIBBS_LN_FUSE=0 timeout 900 python bench.py --no-cpu-baseline --no-strict --no-msa > $OUT/bench_nofold.json 2> $OUT/bench_nofold.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b/bench_nofold.json').read().strip().splitlines()[-1])
print('nofold', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_split_ms_per_iter'])
PY
PGIBBS_LN_FUSE=0 PGIBBS_GEMM_TAIL=0 timeout 900 python bench.py --no-cpu-baseline --no-strict --no-msa > $OUT/bench_nofold_notail.json 2> $OUT/bench_nofold_notail.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b/bench_nofold_notail.json').read().strip().splitlines()[-1])
print('nofold notail', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['time_split_ms_per_iter'])
PY
