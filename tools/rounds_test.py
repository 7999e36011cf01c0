import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v):
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, epi, v, 300, ctypes.byref(ms)))
    return ms.value
for name, N, K, epi in (("qkv", 3840, 1280, 0), ("out", 1280, 1280, 2), ("fc1", 5120, 1280, 1), ("fc2", 1280, 5120, 2)):
    row = []
    for M in (65536 - 256 * 26, 65536 - 256, 65536, 65536 + 256, 66048, 65536 + 256 * 26):
        tiles = (M // 256) * (N // 256)
        row.append("M=%d (%.2f rounds): none %.3f full %.3f" % (M, tiles / 256, run(M, N, K, 0, 30), run(M, N, K, epi, 2)))
    print(name, " | ".join(row))
