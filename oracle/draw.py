"""ORACLE (test infrastructure only -- never imported by the product path).

Token-draw primitive: CPU restatement of `generate_step`
(/root/reference/src/pgen/esm_sampler.py:8-45) with the random source replaced
by the engine's documented counter-based generator ("pg_draw v1").

What follows the reference line by line:
  esm_sampler.py:23      row = out[gen_idx]
  esm_sampler.py:24-25   row / temperature when temperature is not None
  esm_sampler.py:30      sub = row[valid_idx]
  esm_sampler.py:32-38   k = len(sub) if (sample or top_k <= 0 or top_k > len(sub)) else top_k
  esm_sampler.py:40      (vals, ids) = topk(sub, k)  -- descending, ties: lowest position first
  esm_sampler.py:41-43   j ~ Categorical(logits=vals)
  esm_sampler.py:43-45   token = valid_idx[ids[j]]

What cannot follow it: torch's CPU `Categorical.sample()` consumes torch's
global mt19937 (argmax(p/Exp(1))); a data-parallel engine cannot reproduce that
serial stream, so both this oracle and the HIP kernel use the same stateless
generator.  The reference's own tests only pin the *distribution* of the draw
(/root/reference/test/test_esm_sampler.py:185-253); tests/ replays those, and
pins oracle == kernel bit-exactly given identical logits.

pg_draw v1 (every float op is an IEEE-754 binary32 op, no fused multiply-add,
so numpy float32, C with -ffp-contract=off and the HIP kernel agree bitwise):
  s_i   = row[valid_idx[i]] (/ tau)
  rank  = stable descending order of s
  e_r   = pg_exp(s_r - s_0),  r = 0..k-1          (e_0 == 1)
  tot   = ((e_0 + e_1) + e_2) + ...               (sequential, rank order)
  u     = (philox4x32_10(ctr=(row_id, iter, slot, stream), key=(seed_lo, seed_hi))[0] >> 8) * 2^-24
  t     = u * tot
  j     = first r with (e_0 + ... + e_r) > t, else k-1
  token = valid_idx[rank[j]]
pg_exp(x), x <= 0:  z = x*log2(e); 0 if z < -126; n = floor(z); f = z - n;
  p = Horner(c6..c0; f) with separate mul/add;  result = bits(p) + (n << 23).
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85

LOG2E_F32 = np.float32(1.4426950408889634)
EXP2_COEF = [np.float32(float.fromhex(h)) for h in (
    "0x1.000000p+0", "0x1.62e430p-1", "0x1.ebfc40p-3", "0x1.c69f6ap-5",
    "0x1.3c4a8ap-7", "0x1.4ca4cep-10", "0x1.b4d1dep-13")]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al., SC'11).  All args uint32 arrays/scalars."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    for r in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> s32, p0 & mask
        hi1, lo1 = p1 >> s32, p1 & mask
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def uniform24(word):
    """uint32 -> float32 in [0,1) with 24 random bits (exact)."""
    return (np.asarray(word, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def pg_exp(x):
    """Reproducible exp(x) for x <= 0 in binary32 (see module docstring)."""
    x = np.asarray(x, dtype=np.float32)
    z = (x * LOG2E_F32).astype(np.float32)
    flush = z < np.float32(-126.0)
    zc = np.where(flush, np.float32(0.0), z).astype(np.float32)
    n = np.floor(zc).astype(np.float32)
    f = (zc - n).astype(np.float32)
    p = np.full_like(f, EXP2_COEF[6])
    for c in EXP2_COEF[5::-1]:
        p = (p * f).astype(np.float32)
        p = (p + c).astype(np.float32)
    bits = p.view(np.int32) + (n.astype(np.int32) << 23)
    out = bits.view(np.float32)
    return np.where(flush, np.float32(0.0), out).astype(np.float32)


def effective_top_k(n_valid, top_k, sample):
    """esm_sampler.py:32-38."""
    if sample or top_k <= 0 or top_k > n_valid:
        return n_valid
    return top_k


def draw_rows(rows, valid_idx, top_k, sample, temperature, row_id, it, slot, stream, seed):
    """Draw one token per row.

    rows: float32 [n, V] logits (already indexed at gen_idx); row_id/slot: int arrays [n];
    it, stream: ints; seed: 64-bit int.  Returns int32 [n] token ids.
    """
    rows = np.asarray(rows, dtype=np.float32)
    n = rows.shape[0]
    valid_idx = np.asarray(valid_idx, dtype=np.int64)
    nv = len(valid_idx)
    k = effective_top_k(nv, top_k, sample)
    s = rows[:, valid_idx]
    if temperature is not None:
        s = (s / np.float32(temperature)).astype(np.float32)
    # stable descending order == rank by (value desc, position asc)
    order = np.argsort(-s.astype(np.float64), axis=1, kind="stable")
    sv = np.take_along_axis(s, order, axis=1)[:, :k]
    e = pg_exp((sv - sv[:, :1]).astype(np.float32))
    cum = np.empty_like(e)
    acc = e[:, 0].copy()
    cum[:, 0] = acc
    for r in range(1, k):
        acc = (acc + e[:, r]).astype(np.float32)
        cum[:, r] = acc
    w0 = philox4x32_10(np.asarray(row_id, dtype=np.uint32), np.uint32(it), np.asarray(slot, dtype=np.uint32),
                       np.uint32(stream), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)[0]
    u = uniform24(w0)
    t = (u * cum[:, -1]).astype(np.float32)
    hit = cum > t[:, None]
    j = np.where(hit.any(axis=1), hit.argmax(axis=1), k - 1)
    return valid_idx[order[np.arange(n), j]].astype(np.int32)


def generate_step(out, gen_idx, temperature=None, top_k=0, sample=False, valid_idx=None,
                  row_id=0, it=0, slot=0, stream=0, seed=0):
    """Scalar form with the reference's signature (+ the counter that replaces torch's RNG)."""
    out = np.asarray(out, dtype=np.float32)
    if valid_idx is None:
        valid_idx = list(range(out.shape[-1]))
    return int(draw_rows(out[gen_idx][None, :], valid_idx, top_k, sample, temperature,
                         [row_id], it, [slot], stream, seed)[0])
