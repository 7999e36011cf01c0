#!/bin/bash
# A/B of the MSA configs under env settings: each arg is "NAME:VAR=VAL,VAR=VAL"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    timeout 900 python bench_msa.py --config ${CFG:-4} --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin.read().strip().splitlines():
    d=json.loads(ln)
    t=d.get('time_split_ms_per_iter') or d.get('time_split_ms_per_forward')
    print('$name', d['metric'][-10:], 'ms %.2f' % (d.get('ms_per_step') or d.get('ms_per_forward')), 'gemm %.2f attn %.2f ln %.2f' % (t['gemm'], t['attention'], t['layernorm']))
" )
done
