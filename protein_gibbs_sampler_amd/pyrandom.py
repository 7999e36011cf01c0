"""Position selection on the host, bit-exact with the interpreter's global `random`.

The reference picks positions with `random.sample(indexes, num_positions)` once per chain per
iteration (/root/reference/src/pgen/esm_sampler.py:242-246; per MSA row in
esm_msa_sampler.py:272-279) and shuffles with `random.shuffle` (esm_msa_sampler.py:129).  Selection
never depends on the logits, so the whole [iterations x chains x P] table is produced up front by
the native CPython-exact Mersenne Twister (C ABI pg_pyrandom_*): the interpreter's RNG state is
moved in, advanced natively, and moved back, leaving `random` exactly where the reference would.
"""
import ctypes
import random

import numpy as np

from . import _lib


class NativePyRandom:
    def __init__(self):
        self._L = _lib.lib()
        self._h = self._L.pg_pyrandom_create()
        if not self._h:
            raise MemoryError("pg_pyrandom_create failed")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.pg_pyrandom_destroy(h)

    # ---- state interchange with random.getstate()/setstate() ----
    def setstate(self, state):
        version, internal, _gauss = state
        if version != 3 or len(internal) != 625:
            raise ValueError("unsupported random state")
        mt = np.asarray(internal[:624], dtype=np.uint32)
        _lib.check(self._L.pg_pyrandom_setstate(self._h, mt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), int(internal[624])))

    def getstate(self, gauss_next=None):
        mt = np.empty(624, dtype=np.uint32)
        idx = ctypes.c_int(0)
        _lib.check(self._L.pg_pyrandom_getstate(self._h, mt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ctypes.byref(idx)))
        return (3, tuple(int(v) for v in mt) + (idx.value,), gauss_next)

    def seed(self, n):
        n = abs(int(n))
        words = []
        while n:
            words.append(n & 0xFFFFFFFF)
            n >>= 32
        key = np.asarray(words or [0], dtype=np.uint32)
        _lib.check(self._L.pg_pyrandom_seed(self._h, key.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(key)))

    # ---- draws ----
    def sample_table(self, population, k, n_rows):
        pop = np.ascontiguousarray(population, dtype=np.int32)
        if not 0 <= k <= len(pop):
            raise ValueError("Sample larger than population or is negative")
        out = np.empty((n_rows, k), dtype=np.int32)
        _lib.check(self._L.pg_pyrandom_sample_table(self._h, pop.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(pop), k,
                                                    n_rows, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return out

    def sample(self, population, k):
        return self.sample_table(population, k, 1)[0].tolist()

    def shuffle(self, x):
        a = np.ascontiguousarray(x, dtype=np.int32)
        _lib.check(self._L.pg_pyrandom_shuffle(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(a)))
        x[:] = a.tolist()

    def choices(self, n, k):
        out = np.empty(k, dtype=np.int32)
        _lib.check(self._L.pg_pyrandom_choices(self._h, n, k, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return out.tolist()

    def random(self):
        return self._L.pg_pyrandom_random(self._h)

    def getrandbits(self, k):
        return self._L.pg_pyrandom_getrandbits32(self._h, k)


_shared = None


def _native():
    global _shared
    if _shared is None:
        _shared = NativePyRandom()
    return _shared


def global_sample_table(population, k, n_rows):
    """== [random.sample(population, k) for _ in range(n_rows)] on the interpreter's global RNG."""
    r = _native()
    st = random.getstate()
    r.setstate(st)
    out = r.sample_table(population, k, n_rows)
    random.setstate(r.getstate(st[2]))
    return out


def global_shuffle(x):
    """== random.shuffle(x) on the interpreter's global RNG (x: list of ints)."""
    r = _native()
    st = random.getstate()
    r.setstate(st)
    r.shuffle(x)
    random.setstate(r.getstate(st[2]))
