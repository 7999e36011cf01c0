#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r04j; mkdir -p $O
timeout 300 ./build/grid_barrier_bench > $O/barrier.txt 2>&1; cat $O/barrier.txt
for ct in 1 0; do for gr in 1 0; do PGIBBS_CHAIN_TRUNK=$ct PGIBBS_GRAPH=$gr timeout 300 python tools/probes/chain_trunk_debug.py $O/dbg_ct${ct}_g${gr}.npz 2>&1 | tail -2; done; done
python - <<'PY'
import numpy as np
O="gpurun_out/r04j/"
ref=np.load(O+"dbg_ct0_g0.npz")
for name in ("dbg_ct0_g1","dbg_ct1_g0","dbg_ct1_g1"):
    d=np.load(O+name+".npz")
    for k in ref.files:
        a,b=ref[k],d[k]
        same = np.array_equal(a.view(np.uint32) if a.dtype==np.float32 else a, b.view(np.uint32) if b.dtype==np.float32 else b)
        extra=""
        if not same and a.dtype==np.float32:
            diff=np.abs(a-b); extra=" max|d|=%.4g first differing iteration %s"%(diff.max(), np.argwhere(diff.reshape(diff.shape[0],-1).max(1)>0)[:3].ravel())
        print(name,k,"same" if same else "DIFF"+extra)
PY
