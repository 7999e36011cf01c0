"""ORACLE (test infrastructure only -- never imported by the product path).

CPython-exact restatement of the `random` module calls the reference makes on
its Gibbs hot path.  The reference draws positions with the *global* Python
Mersenne Twister:

  * `random.sample(indexes, num_positions)`  -- /root/reference/src/pgen/esm_sampler.py:245,
                                                 esm_msa_sampler.py:277
  * `random.choices(seed_seq, k=batch_size)`  -- esm_sampler.py:112
  * `random.shuffle(positions)`               -- esm_msa_sampler.py:129

The algorithm lives in CPython itself (Modules/_randommodule.c, Lib/random.py;
CPython 3.9/3.10 are identical for these calls), restated here from its
published description:

  seed(int n): key = little-endian 32-bit words of |n| -> MT19937 init_by_array
  getrandbits(k<=32) = genrand_uint32() >> (32-k)
  _randbelow(n): k = n.bit_length(); r = getrandbits(k); while r >= n: redraw
  random() = ((a>>5)*2**26 + (b>>6)) / 2**53   (a, b two successive uint32)
  sample(pop, k): pool path if n <= setsize else set path
  shuffle(x): for i in reversed(range(1, len(x))): j = randbelow(i+1); swap
  choices(pop, k): pop[floor(random()*n)] k times

Pinned by tests/test_oracle_pyrandom.py against the interpreter's own `random`
module (the very module the reference calls) for seeds x shapes, including
getstate()/setstate() interchange.
"""
from math import ceil, log

N = 624
M = 397
MATRIX_A = 0x9908B0DF
UPPER = 0x80000000
LOWER = 0x7FFFFFFF
MASK32 = 0xFFFFFFFF


class PyRandom:
    """Bit-exact model of CPython's `random.Random` for the calls listed above."""

    def __init__(self, seed=None):
        self.mt = [0] * N
        self.idx = N
        if seed is not None:
            self.seed(seed)

    # ---- state plumbing -------------------------------------------------
    def _init_genrand(self, s):
        mt = self.mt
        mt[0] = s & MASK32
        for i in range(1, N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & MASK32
        self.idx = N

    def _init_by_array(self, key):
        self._init_genrand(19650218)
        mt = self.mt
        i, j = 1, 0
        klen = len(key)
        for _ in range(max(N, klen)):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & MASK32
            i += 1
            j += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
            if j >= klen:
                j = 0
        for _ in range(N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & MASK32
            i += 1
            if i >= N:
                mt[0] = mt[N - 1]
                i = 1
        mt[0] = 0x80000000

    def seed(self, n):
        """random.seed(int): key = 32-bit little-endian words of abs(n)."""
        n = abs(int(n))
        key = []
        while n:
            key.append(n & MASK32)
            n >>= 32
        if not key:
            key = [0]
        self._init_by_array(key)

    def getstate(self):
        """Same tuple shape as random.getstate(): (3, (mt..., idx), None)."""
        return (3, tuple(self.mt) + (self.idx,), None)

    def setstate(self, state):
        version, internal, _gauss = state
        assert version == 3
        self.mt = list(internal[:N])
        self.idx = internal[N]

    # ---- generator ------------------------------------------------------
    def _genrand_uint32(self):
        mt = self.mt
        if self.idx >= N:
            for kk in range(N - M):
                y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER)
                mt[kk] = mt[kk + M] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
            for kk in range(N - M, N - 1):
                y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER)
                mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
            y = (mt[N - 1] & UPPER) | (mt[0] & LOWER)
            mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (MATRIX_A if (y & 1) else 0)
            self.idx = 0
        y = mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & MASK32

    def getrandbits(self, k):
        assert 0 < k <= 32, "only the <=32-bit path is on the hot path"
        return self._genrand_uint32() >> (32 - k)

    def random(self):
        a = self._genrand_uint32() >> 5
        b = self._genrand_uint32() >> 6
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)

    def _randbelow(self, n):
        if n == 0:
            raise ValueError("empty range")
        k = n.bit_length()
        r = self.getrandbits(k)
        while r >= n:
            r = self.getrandbits(k)
        return r

    # ---- the three calls the reference makes ----------------------------
    def sample(self, population, k):
        population = list(population) if not isinstance(population, (list, range, tuple)) else population
        n = len(population)
        if not 0 <= k <= n:
            raise ValueError("Sample larger than population or is negative")
        result = [None] * k
        setsize = 21
        if k > 5:
            setsize += 4 ** ceil(log(k * 3, 4))
        if n <= setsize:
            pool = list(population)
            for i in range(k):
                j = self._randbelow(n - i)
                result[i] = pool[j]
                pool[j] = pool[n - i - 1]
        else:
            selected = set()
            for i in range(k):
                j = self._randbelow(n)
                while j in selected:
                    j = self._randbelow(n)
                selected.add(j)
                result[i] = population[j]
        return result

    def shuffle(self, x):
        for i in reversed(range(1, len(x))):
            j = self._randbelow(i + 1)
            x[i], x[j] = x[j], x[i]

    def choices(self, population, k=1):
        n = len(population)
        from math import floor
        return [population[floor(self.random() * n)] for _ in range(k)]
