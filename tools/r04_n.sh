#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04n; mkdir -p $O
PGIBBS_CHAIN_TRUNK=0 python tools/probes/chain_trunk_bits.py $O/ref.npz 2>&1 | grep -v amdgpu.ids
PGIBBS_CHAIN_TRUNK=1 python tools/probes/chain_trunk_bits.py $O/sc1.npz 2>&1 | grep -v amdgpu.ids
PGIBBS_CHAIN_TRUNK=1 PGIBBS_LIB_PATH=$PWD/build/libpgibbs_aux17.so python tools/probes/chain_trunk_bits.py $O/sys.npz 2>&1 | grep -v amdgpu.ids
python - <<'PY'
import numpy as np
O="gpurun_out/r04n/"
ref=np.load(O+"ref.npz")
for name in ("sc1","sys"):
    d=np.load(O+name+".npz"); bad=0
    for k in ref.files:
        if not np.array_equal(ref[k].view(np.uint32), d[k].view(np.uint32)):
            bad+=1; print(name,k,"DIFF max|d|=%.3g, %d of %d values"%(np.abs(ref[k]-d[k]).max(), (ref[k]!=d[k]).sum(), ref[k].size))
    print(name,"differing arrays:",bad,"of",len(ref.files))
PY
for lib in "" $PWD/build/libpgibbs_aux17.so; do echo "lib=$lib"; PGIBBS_LIB_PATH=$lib PGIBBS_CHAIN_TRUNK=1 timeout 300 python tools/cfg1_probe.py 2>&1 | grep "stream=own" | tail -1; done
