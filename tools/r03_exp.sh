#!/bin/bash
# quick A/B of the config-2 bench under env settings: each arg is "NAME:VAR=VAL,VAR=VAL"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict --no-msa --no-host-entry 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=d['time_split_ms_per_iter']
print('$name', 'ms/step %.2f' % d['ms_per_step'], 'gemm %.2f attn %.2f ln %.2f' % (t['gemm'], t['attention'], t['layernorm']))
" )
done
