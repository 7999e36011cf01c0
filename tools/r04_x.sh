#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04x; mkdir -p $O
python -m pytest tests/test_gpu_msa.py tests/test_gpu_config5_and_protocol.py tests/test_gpu_fullsize_logits.py -x -q 2>&1 | tail -3
for rep in 1 2; do for lib in "" $PWD/build/libpgibbs_old.so; do for c in 4 5; do echo "lib=${lib:-new} config $c"; PGIBBS_LIB_PATH=$lib python bench_msa.py --config $c --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in j.items() if k.startswith('ms_') or k=='time_split_ms_per_iter' or k=='time_split_ms_per_forward'})"; done; done; done 2>&1 | tee $O/ab.txt
