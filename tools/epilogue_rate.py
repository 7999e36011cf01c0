import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
for M in (256, 2048, 4096, 16384, 66048):
    for v in (33, 34):
        ms = ctypes.c_double()
        _lib.check(L.pg_dbg_gemm_bench(0, M, 3840, 1280, 0, v, 20, ctypes.byref(ms)))
        tiles = (M // 256) * 15
        nbytes = M * 3840 * (2 if v == 33 else 8)
        print("M=%6d v%d: %.4f ms  tiles=%5d  %.1f GB/s total  %.1f GB/s per busy CU" % (M, v, ms.value, tiles, nbytes / ms.value / 1e6, nbytes / ms.value / 1e6 / min(tiles, 256)))
