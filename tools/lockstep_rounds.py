"""One LOCKSTEP round of tiles (a launch of <= 256 tiles: every CU starts its tile together, the 32 tiles of an XCD walk K together
and share k-slices through its L2) against the per-round time of the full-size launch, where CUs pick up tiles whenever they finish
and the groups drift apart.  If a lone round -- launch gap, fill and drain included -- is not slower than a steady-state round, rounds
synchronised per XCD would not lose; PMC: FETCH_SIZE of the two (tools/lockstep_rounds.sh)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_gibbs_sampler_amd import _lib
L = _lib.lib()
def run(M, N, K, epi, v, it=200):
    ms = ctypes.c_double()
    _lib.check(L.pg_dbg_gemm_bench(0, M, N, K, epi, v, it, ctypes.byref(ms)))
    return ms.value * 1e3
for name, N, K, epi, v, m_one in (("qkv", 3840, 1280, 0, 80, 17), ("fc1", 5120, 1280, 1, 80, 12), ("out", 1280, 1280, 2, 20, 51), ("fc2", 1280, 5120, 2, 20, 51)):
    tn = N // 256
    for rep in range(2):
        one = run(m_one * 256, N, K, epi, v)
        two = run(2 * m_one * 256, N, K, epi, v)
        full_m = 258
        full = run(full_m * 256, N, K, epi, v)
        rounds = full_m * tn / 256.0
        print("%s: one round (%d tiles) %.1f us | two rounds %.1f us | full size %.1f us = %.2f rounds x %.1f us | lone round / steady round = %.2f"
              % (name, m_one * tn, one, two, full, rounds, full / rounds, one * (256.0 / (m_one * tn)) / (full / rounds)), flush=True)
