#!/bin/bash
# round 4, run C: fp16-operand mode -- kernel tests, full-size logit figures, bench with the fp16 leg; plus the new batch tests
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_gpu_fp16_mode.py -x -q -s > $O/test_fp16.txt 2>&1; tail -15 $O/test_fp16.txt
python -m pytest tests/test_gpu_fullsize_logits.py -x -q -s -k "fp16 or bf16" > $O/test_fullsize.txt 2>&1; grep -E "^\[|passed|failed|Error" $O/test_fullsize.txt
python -m pytest tests/test_gpu_config5_and_protocol.py -x -q -k "default_pgen" > $O/test_cfg5.txt 2>&1; tail -3 $O/test_cfg5.txt
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_msa.py -x -q > $O/test_misc.txt 2>&1; tail -3 $O/test_misc.txt
python bench.py --steps 10 --warmup 3 --no-msa > $O/bench.json 2> $O/bench.err; tail -c 6000 $O/bench.json
