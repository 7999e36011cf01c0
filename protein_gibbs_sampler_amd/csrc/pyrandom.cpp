// CPython-exact Mersenne Twister for host-side position selection (C ABI: pg_pyrandom_*).
//
// Replaces the reference's calls into the interpreter's global RNG on the Gibbs hot path:
//   random.sample   /root/reference/src/pgen/esm_sampler.py:245, esm_msa_sampler.py:277
//   random.choices  esm_sampler.py:112          random.shuffle  esm_msa_sampler.py:129
// Algorithm: CPython Modules/_randommodule.c (MT19937, init_by_array) and Lib/random.py
// (_randbelow_with_getrandbits, sample's pool/set split, shuffle, choices) -- SURVEY.md Appendix B.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <unordered_set>
#include <vector>

#include "../../include/pgibbs.h"
#include "pg_common.h"

struct pg_pyrandom {
  uint32_t mt[624];
  int idx;
};

namespace {
constexpr int N = 624, M = 397;

void init_genrand(pg_pyrandom* r, uint32_t s) {
  r->mt[0] = s;
  for (int i = 1; i < N; ++i) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->idx = N;
}

void init_by_array(pg_pyrandom* r, const uint32_t* key, int klen) {
  init_genrand(r, 19650218u);
  uint32_t* mt = r->mt;
  int i = 1, j = 0;
  for (int k = (N > klen ? N : klen); k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    ++i; ++j;
    if (i >= N) { mt[0] = mt[N - 1]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (int k = N - 1; k; --k) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    ++i;
    if (i >= N) { mt[0] = mt[N - 1]; i = 1; }
  }
  mt[0] = 0x80000000u;
}

inline uint32_t genrand(pg_pyrandom* r) {
  uint32_t* mt = r->mt;
  if (r->idx >= N) {
    int kk;
    for (kk = 0; kk < N - M; ++kk) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < N - 1; ++kk) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    uint32_t y = (mt[N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r->idx = 0;
  }
  uint32_t y = mt[r->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

inline int bit_length(uint32_t n) { return n ? 32 - __builtin_clz(n) : 0; }

inline uint32_t randbelow(pg_pyrandom* r, uint32_t n) {
  const int k = bit_length(n);
  uint32_t v = genrand(r) >> (32 - k);
  while (v >= n) v = genrand(r) >> (32 - k);
  return v;
}

// Lib/random.py sample(): setsize = 21 (+ 4 ** ceil(log(3k, 4)) when k > 5); n <= setsize -> pool path
inline int64_t sample_setsize(int k) {
  int64_t setsize = 21;
  if (k > 5) {
    const int e = (int)ceil(log((double)k * 3.0) / log(4.0));  // math.log(x, 4) == log(x)/log(4) in CPython
    setsize += (int64_t)1 << (2 * e);
  }
  return setsize;
}

void sample_one(pg_pyrandom* r, const int32_t* pop, int n, int k, int32_t* out, std::vector<int32_t>& pool,
                std::unordered_set<uint32_t>& sel) {
  if ((int64_t)n <= sample_setsize(k)) {
    pool.assign(pop, pop + n);
    for (int i = 0; i < k; ++i) {
      const uint32_t j = randbelow(r, (uint32_t)(n - i));
      out[i] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  } else {
    sel.clear();
    for (int i = 0; i < k; ++i) {
      uint32_t j = randbelow(r, (uint32_t)n);
      while (sel.count(j)) j = randbelow(r, (uint32_t)n);
      sel.insert(j);
      out[i] = pop[j];
    }
  }
}
}  // namespace

extern "C" {

pg_pyrandom* pg_pyrandom_create(void) {
  pg_pyrandom* r = new pg_pyrandom;
  const uint32_t key = 0;
  init_by_array(r, &key, 1);
  return r;
}
void pg_pyrandom_destroy(pg_pyrandom* r) { delete r; }

int pg_pyrandom_seed(pg_pyrandom* r, const uint32_t* key_words, int n_words) {
  if (!r || !key_words || n_words < 1) return pg::fail(PG_ERR_INVALID, "pyrandom_seed: need >= 1 key word");
  init_by_array(r, key_words, n_words);
  return PG_OK;
}
int pg_pyrandom_setstate(pg_pyrandom* r, const uint32_t* mt624, int index) {
  if (!r || !mt624 || index < 0 || index > N) return pg::fail(PG_ERR_INVALID, "pyrandom_setstate: bad state");
  memcpy(r->mt, mt624, sizeof(r->mt));
  r->idx = index;
  return PG_OK;
}
int pg_pyrandom_getstate(const pg_pyrandom* r, uint32_t* mt624, int* index) {
  if (!r || !mt624 || !index) return pg::fail(PG_ERR_INVALID, "pyrandom_getstate: null");
  memcpy(mt624, r->mt, sizeof(r->mt));
  *index = r->idx;
  return PG_OK;
}
uint32_t pg_pyrandom_getrandbits32(pg_pyrandom* r, int k) {
  if (k < 1) k = 1;
  if (k > 32) k = 32;
  return genrand(r) >> (32 - k);
}
double pg_pyrandom_random(pg_pyrandom* r) {
  const uint32_t a = genrand(r) >> 5, b = genrand(r) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}
int pg_pyrandom_sample(pg_pyrandom* r, const int32_t* population, int n, int k, int32_t* out) {
  return pg_pyrandom_sample_table(r, population, n, k, 1, out);
}
int pg_pyrandom_sample_table(pg_pyrandom* r, const int32_t* population, int n, int k, int64_t n_rows, int32_t* out) {
  if (!r || n < 0 || k < 0 || k > n) return pg::fail(PG_ERR_INVALID, "Sample larger than population or is negative");
  if (k == 0 || n_rows == 0) return PG_OK;
  if (!population || !out) return pg::fail(PG_ERR_INVALID, "pyrandom_sample: null buffer");
  std::vector<int32_t> pool;
  std::unordered_set<uint32_t> sel;
  for (int64_t row = 0; row < n_rows; ++row) sample_one(r, population, n, k, out + row * k, pool, sel);
  return PG_OK;
}
int pg_pyrandom_shuffle(pg_pyrandom* r, int32_t* x, int n) {
  if (!r || (n > 0 && !x)) return pg::fail(PG_ERR_INVALID, "pyrandom_shuffle: null");
  for (int i = n - 1; i >= 1; --i) {
    const uint32_t j = randbelow(r, (uint32_t)(i + 1));
    const int32_t t = x[i];
    x[i] = x[j];
    x[j] = t;
  }
  return PG_OK;
}
int pg_pyrandom_choices(pg_pyrandom* r, int n, int k, int32_t* out) {
  if (!r || n < 1 || k < 0 || (k > 0 && !out)) return pg::fail(PG_ERR_INVALID, "pyrandom_choices: bad arguments");
  for (int i = 0; i < k; ++i) out[i] = (int32_t)floor(pg_pyrandom_random(r) * n);
  return PG_OK;
}

}  // extern "C"
