#!/bin/bash
# round 4, run H: fused column QKV + attention (gemm_colattn.hip): bit-identity, MSA parity tests, configs 4 and 5 fused vs unfused
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests/test_gpu_msa.py -x -q -k "fused_column" > $O/test_colfuse.txt 2>&1; tail -5 $O/test_colfuse.txt
python -m pytest tests/test_gpu_msa.py tests/test_gpu_fullsize_logits.py -x -q -k "msa" > $O/test_msa.txt 2>&1; tail -3 $O/test_msa.txt
for f in 0 1 0 1; do
  PGIBBS_MSA_COLFUSE=$f python bench_msa.py --config 4 --steps 5 > $O/msa4_$f.json 2>/dev/null
  python -c "import json; d=json.loads(open('$O/msa4_$f.json').read().strip().splitlines()[-1]); print('cfg4 fuse=$f', round(d['ms_per_step'],2), d['time_split_ms_per_iter'])"
done
for f in 0 1; do
  PGIBBS_MSA_COLFUSE=$f python bench_msa.py --config 5 > $O/msa5_$f.json 2>/dev/null
  python -c "import json; d=json.loads(open('$O/msa5_$f.json').read().strip().splitlines()[-1]); print('cfg5 fuse=$f', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_forward','ms_per_template_forward','value','time_split_ms_per_forward')})"
done
