// Strict-precision attention (PG_PREC_FP32): the same softmax(q k^T) v per (sequence, head) as attention.hip from
// fp32 q, k, v, for the 1e-3 logit-parity mode.
//
// attention_split_kernel (the default): the MFMA formulation of attention.hip's attention_long_kernel with every
// operand split into a bf16 (hi, lo) pair in registers -- S = ql.kh + qh.kl + qh.kh and O = vl.ph + vh.pl + vh.ph, fp32
// accumulation, small terms first -- so q, k, v and P keep ~16 mantissa bits (the dropped lo.lo terms are ~2^-17
// relative).  One workgroup = (sequence, head, 64 queries); K and V tiles of MAXKB*16 keys are split while they are
// staged (four LDS tiles: Kh, Kl, Vh, Vl); online softmax across tiles; exp2 on fp32 scores.
//
// attention_f32_kernel (PGIBBS_ATTN_F32=valu): the all-VALU fp32 form, kept as an independent cross-check.  One
// 64-lane workgroup = 64 queries of one (sequence, head); a thread owns one query: q[64] and o[64] in registers, K/V
// tiles of 64 keys staged in LDS as fp32 (all lanes read the same K/V row -> LDS broadcast), online softmax.
#include <cstdlib>

#include "kernels.h"

PG_OPS_BEGIN

// One thread's 64 context values of head h -> bf16 into the token's context row.  split_d != 0: the row is the strict mode's
// split operand (3 * |split_d| values, groups of 32 columns [lo | hi | hi]; elementwise.hip store_row_bf16); split_d < 0: without
// the duplicate hi block -- the out-projection that reads the rows is the fused three-product kernel (gemm_split3_fused).
__device__ __forceinline__ void store_ctx64(const float (&o)[64], float inv, bf16_t* row, int h, int split_d) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float a = o[4 * i] * inv, b = o[4 * i + 1] * inv, c = o[4 * i + 2] * inv, d = o[4 * i + 3] * inv;
    uint2 p, r;
    p.x = pack_op2(a, b);
    p.y = pack_op2(c, d);
    if (!split_d) {
      ((uint2*)(row + h * 64))[i] = p;
    } else {
      r.x = pack_op2(a - op16_to_f32((bf16_t)(p.x & 0xffff)), b - op16_to_f32((bf16_t)(p.x >> 16)));
      r.y = pack_op2(c - op16_to_f32((bf16_t)(p.y & 0xffff)), d - op16_to_f32((bf16_t)(p.y >> 16)));
      bf16_t* g = row + (2 * h + (i >> 3)) * 96 + (i & 7) * 4;       // columns h*64 + 4i .. +3
      *(uint2*)g = r;
      *(uint2*)(g + 32) = p;
      if (split_d > 0) *(uint2*)(g + 64) = p;
    }
  }
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef short v4s __attribute__((ext_vector_type(4)));

// 8 fp32 -> 8 bf16 hi (round to nearest even) + 8 bf16 lo = bf16(v - hi)
__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
  hi.x = pack_op2(a.x, a.y); hi.y = pack_op2(a.z, a.w); hi.z = pack_op2(b.x, b.y); hi.w = pack_op2(b.z, b.w);
  lo.x = pack_op2(a.x - __uint_as_float(hi.x << 16), a.y - __uint_as_float(hi.x & 0xffff0000u));
  lo.y = pack_op2(a.z - __uint_as_float(hi.y << 16), a.w - __uint_as_float(hi.y & 0xffff0000u));
  lo.z = pack_op2(b.x - __uint_as_float(hi.z << 16), b.y - __uint_as_float(hi.z & 0xffff0000u));
  lo.w = pack_op2(b.z - __uint_as_float(hi.w << 16), b.w - __uint_as_float(hi.w & 0xffff0000u));
}

// Building blocks shared by the three split-bf16 MFMA kernels (full attention, tied-row scores, tied-row apply).
// Fragment geometry as in attention.hip: fr = lane & 15 is the wave's query (MFMA column), fq = lane >> 4; a score
// block st[kb][r] = S[query fr][key kb*16 + fq*4 + r]; O^T blocks o[db][r] = O[query fr][d = db*16 + fq*4 + r].
template <int MAXKB>
struct SplitAttn {
  static constexpr int tpad = MAXKB * 16, nkc = MAXKB / 2;
  static_assert(MAXKB % 2 == 0, "two 16-key blocks per 32-wide PV step");
  static constexpr float LOG2E = 1.44269504088896341f;

  // rows 0 .. tpad-1 of 64 fp32 at src + row*ld (rows >= n_valid read as zero) -> (hi, lo) bf16 tiles; a tile row is
  // 128 B with its 16-B chunks XOR-swizzled: row*128 + ((c ^ (row & 7)) << 4).  All global loads in flight first.
  // extra (optional): 64 fp32 that stand in for row n_valid (ESM-1: the head's bias_k / bias_v behind the last token)
  template <bool EXTRA = false>
  static __device__ __forceinline__ void stage(const float* __restrict__ src, size_t ld, int n_valid, char* Xh, char* Xl, int tid,
                                               const float* __restrict__ extra = nullptr) {
    constexpr int NIT = (tpad * 8 + 255) / 256;      // one item = 8 d of one key
    float4 r0[NIT], r1[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256, row = i >> 3, c = i & 7;
      r0[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      r1[it] = r0[it];
      if (i < tpad * 8 && row < n_valid) {
        const float4* p = (const float4*)(src + (size_t)row * ld + c * 8);
        r0[it] = p[0];
        r1[it] = p[1];
      } else if (EXTRA && row == n_valid && i < tpad * 8) {
        const float4* p = (const float4*)(extra + c * 8);
        r0[it] = p[0];
        r1[it] = p[1];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256, row = i >> 3, c = i & 7;
      if (i < tpad * 8) {
        uint4 hi, lo;
        split8(r0[it], r1[it], hi, lo);
        const int a = row * 128 + ((c ^ (row & 7)) << 4);
        *(uint4*)(Xh + a) = hi;
        *(uint4*)(Xl + a) = lo;
      }
    }
  }

  // Q fragments (MFMA B operand) of the query whose 64 fp32 start at qrow: d = kk*32 + fq*8 .. +7
  static __device__ __forceinline__ void load_q(const float* __restrict__ qrow, int fq, bf16x8 (&qh)[2], bf16x8 (&ql)[2]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float4* p = (const float4*)(qrow + kk * 32 + fq * 8);
      uint4 hi, lo;
      split8(p[0], p[1], hi, lo);
      qh[kk] = __builtin_bit_cast(bf16x8, hi);
      ql[kk] = __builtin_bit_cast(bf16x8, lo);
    }
  }

  // st[kb] += (K tile rows kb*16 .. +15) . q :  kl.qh + kh.ql + kh.qh
  static __device__ __forceinline__ void qk(const char* Kh, const char* Kl, const bf16x8 (&qh)[2], const bf16x8 (&ql)[2],
                                            f32x4 (&st)[MAXKB], int fr, int fq) {
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
      f32x4 a = st[kb];
      const int krow = kb * 16 + fr;
      bf16x8 kh[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ad = krow * 128 + (((kk * 4 + fq) ^ (krow & 7)) << 4);
        kh[kk] = *(const bf16x8*)(Kh + ad);
        a = mfma_op16(*(const bf16x8*)(Kl + ad), qh[kk], a);
        a = mfma_op16(kh[kk], ql[kk], a);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) a = mfma_op16(kh[kk], qh[kk], a);
      st[kb] = a;
    }
  }

  // One online-softmax step over the tile's scores st (masked entries hold -3e38), then O += V^T P^T with P and V split.
  // K-slot (fq*8 + j) of 32-key chunk c <-> key (2c + (j>>2))*16 + fq*4 + (j&3): exactly the order the lane holds P in.
  static __device__ __forceinline__ void softmax_pv(f32x4 (&st)[MAXKB], f32x4 (&o)[4], float& m, float& l, const char* Vh,
                                                    const char* Vl, int fr, int fq) {
    float tmax = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, st[kb][r]);
    tmax = rows4_max(tmax);
    const float mn = fmaxf(m, tmax);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E);
    const float mneg = -mn * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = st[kb][r] > -1.0e38f ? __builtin_amdgcn_exp2f(fmaf(st[kb][r], LOG2E, mneg)) : 0.f;
        st[kb][r] = e;
        psum += e;
      }
    psum = rows4_sum(psum);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
    }
#pragma unroll
    for (int c = 0; c < nkc; ++c) {
      const f32x4 p0 = st[2 * c], p1 = st[2 * c + 1];
      uint4 ph, pl;
      split8(make_float4(p0[0], p0[1], p0[2], p0[3]), make_float4(p1[0], p1[1], p1[2], p1[3]), ph, pl);
      const bf16x8 pfh = __builtin_bit_cast(bf16x8, ph), pfl = __builtin_bit_cast(bf16x8, pl);
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        union { bf16x8 v; uint2 h2[2]; } vh, vl;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {           // transposed LDS read (attention.hip): lane fr gets V[key0..key0+3][d = db*16 + fr]
          const int krow = (2 * c + hh) * 16 + fq * 4 + (fr >> 2);
          const int dcol = db * 16 + (fr & 3) * 4;
          const int ad = krow * 128 + (((dcol >> 3) ^ (krow & 7)) << 4) + ((dcol >> 2) & 1) * 8;
          vh.h2[hh] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(
                                                    (__attribute__((address_space(3))) char*)(Vh + ad))));
          vl.h2[hh] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(
                                                    (__attribute__((address_space(3))) char*)(Vl + ad))));
        }
        o[db] = mfma_op16(vl.v, pfh, o[db]);
        o[db] = mfma_op16(vh.v, pfl, o[db]);
        o[db] = mfma_op16(vh.v, pfh, o[db]);
      }
    }
  }

  // row = the query's context row; the lane holds columns h*64 + db*16 + fq*4 .. +3.  bf16, or the strict mode's split operand
  // row (groups of 32 columns [lo | hi | hi])
  static __device__ __forceinline__ void store_ctx(const f32x4 (&o)[4], float l, bf16_t* row, int h, int fq, int split_d) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;     // every key masked (an all-<pad> sequence): zero context, not NaN
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const float a = o[db][0] * inv, b = o[db][1] * inv, c = o[db][2] * inv, d = o[db][3] * inv;
      uint2 p, r;
      p.x = pack_op2(a, b);
      p.y = pack_op2(c, d);
      if (!split_d) {
        *(uint2*)(row + h * 64 + db * 16 + fq * 4) = p;
      } else {
        r.x = pack_op2(a - __uint_as_float(p.x << 16), b - __uint_as_float(p.x & 0xffff0000u));
        r.y = pack_op2(c - __uint_as_float(p.y << 16), d - __uint_as_float(p.y & 0xffff0000u));
        bf16_t* g = row + (2 * h + (db >> 1)) * 96 + (db & 1) * 16 + fq * 4;
        *(uint2*)g = r;
        *(uint2*)(g + 32) = p;
        if (split_d > 0) *(uint2*)(g + 64) = p;
      }
    }
  }
};

// ---- full attention: one workgroup = (sequence, head, 64 * NQB queries); wave w owns the 16-query blocks w, w + 4, ... of the
// chunk.  The K / V tiles are read from HBM / L2 and split ONCE per workgroup and tile and every query block of the chunk runs
// against them, its online-softmax state (O, m, l: 18 registers) carried across the tiles: at T = 258 one workgroup per
// (sequence, head) instead of five that each re-staged and re-split the same 132 KB of fp32 K and V (round 3: 28 ms of the strict
// config-2 iteration were this kernel, 4x the bf16 mode's, most of it the 5x redundant fp32 tile traffic).  The per-query
// arithmetic (tile order, products, rounding points) is unchanged: identical bits.
template <int MAXKB, int NQB, bool BIASKV = false, bool PADMASK = false>
__global__ __launch_bounds__(256, 2) void attention_split_kernel(
    const float* __restrict__ qkv, bf16_t* __restrict__ ctx, int split_d, int T, int H, int ld_qkv_, int ld_ctx_, int k_off,
    int v_off, SeqLayout sl, int n_qchunk, const int32_t* __restrict__ key_tok, int pad_idx, const float* __restrict__ bias_kv) {
  using A = SplitAttn<MAXKB>;
  constexpr int tpad = A::tpad;
  __shared__ __attribute__((aligned(16))) char smem[4 * tpad * 128];
  char *Kh = smem, *Kl = smem + tpad * 128, *Vh = smem + 2 * tpad * 128, *Vl = smem + 3 * tpad * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qc = blockIdx.x % n_qchunk, sh = blockIdx.x / n_qchunk;
  const int seq = sh / H, h = sh % H;
  const size_t row0 = (size_t)(seq / sl.inner_count) * sl.outer_rows + (size_t)(seq % sl.inner_count) * sl.inner_rows;
  const size_t ld_qkv = (size_t)ld_qkv_ * sl.row_step, ld_ctx = (size_t)ld_ctx_ * sl.row_step;
  const float* base = qkv + row0 * ld_qkv_ + h * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int qbase = qc * (64 * NQB) + wave * 16;     // block j of this wave: queries qbase + j*64 .. +15
  f32x4 o[NQB][4];
  float m[NQB], l[NQB];
#pragma unroll
  for (int j = 0; j < NQB; ++j) {
#pragma unroll
    for (int db = 0; db < 4; ++db) o[j][db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[j] = -3.0e38f;
    l[j] = 0.f;
  }

  // ESM-1 (add_bias_kv): key T = this head's bias_k / bias_v -- one more key, never masked (attention.hip)
  const int Tk = T + (BIASKV ? 1 : 0);
  const float* bk = BIASKV ? bias_kv + h * 64 : nullptr;
  const float* bv = BIASKV ? bias_kv + (H + h) * 64 : nullptr;
  for (int k0 = 0; k0 < Tk; k0 += tpad) {
    __syncthreads();
    A::template stage<BIASKV>(base + (size_t)k0 * ld_qkv + k_off, ld_qkv, T - k0, Kh, Kl, tid, bk);
    A::template stage<BIASKV>(base + (size_t)k0 * ld_qkv + v_off, ld_qkv, T - k0, Vh, Vl, tid, bv);
    // the tile's <pad> flags once per tile and wave, as wave-uniform bit masks (read per key, per query block and per
    // tile inside the score loop they were the pattern that made the ragged bf16 attention 5x slower before its flags were staged;
    // here the four fp32 tiles already fill the LDS two workgroups per CU may hold, so the flags live in scalar registers)
    unsigned long long pm[(tpad + 63) / 64];
    if (PADMASK) {
#pragma unroll
      for (int w = 0; w < (tpad + 63) / 64; ++w) {
        const int key = w * 64 + lane;
        pm[w] = __ballot(key < tpad && k0 + key < T && key_tok[row0 + (size_t)(k0 + key) * sl.row_step] == pad_idx);
      }
    }
    __syncthreads();
    const int tl = Tk - k0 - fq * 4;             // key k0 + kb*16 + fq*4 + r is padding iff kb*16 + r >= tl
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
      const int q0 = qbase + j * 64;
      if (q0 >= T) continue;                       // wave-uniform
      bf16x8 qh[2], ql[2];
      A::load_q(base + (size_t)(q0 + fr < T ? q0 + fr : T - 1) * ld_qkv, fq, qh, ql);
      f32x4 st[MAXKB];
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      A::qk(Kh, Kl, qh, ql, st, fr, fq);
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb * 16 + r >= tl) st[kb][r] = -3.0e38f;
      if (PADMASK) {      // <pad> keys of a ragged batch: token of key t at key_tok[row0 + t * row_step]; -inf for chains, fair-esm's
                          // finite -10000 for the MSA Transformer's column attention (attention.hip)
        const float fill = sl.row_step == 1 ? -3.0e38f : -10000.0f;
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
          const uint32_t f4 = (uint32_t)(pm[kb >> 2] >> ((kb & 3) * 16 + fq * 4)) & 0xfu;      // keys kb*16 + fq*4 .. +3 of the tile
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((f4 >> r) & 1u) st[kb][r] = fill;
        }
      }
      A::softmax_pv(st, o[j], m[j], l[j], Vh, Vl, fr, fq);
    }
  }
#pragma unroll
  for (int j = 0; j < NQB; ++j) {
    const int q = qbase + j * 64 + fr;
    if (qbase + j * 64 < T && q < T) A::store_ctx(o[j], l[j], ctx + row0 * ld_ctx_ + (size_t)q * ld_ctx, h, fq, split_d);
  }
}

// ---- tied row attention (MSA), step 1: S[b,h][i][j] = scale * sum_r q_r[i] . k_r[j] ---------------------------------
// One workgroup = (msa b, head h, 64 queries, one key tile); the score blocks stay in the MFMA accumulators while the
// R alignment rows stream through the K tile.  S rows are ldS floats (C rounded up to 4) for 16-byte accesses.
template <int MAXKB>
__global__ __launch_bounds__(256, 2) void msa_row_scores_split_kernel(const float* __restrict__ qkv, float* __restrict__ S, int R,
                                                                      int C, int H, int ld_qkv, int k_off, float scale,
                                                                      int n_qchunk, int n_ktile, int ldS,
                                                                      const int32_t* __restrict__ tok, int pad_idx) {
  // tok (optional: a ragged list of MSAs padded to one [B][R][C] tensor, esm_msa_sampler.py:341): fair-esm's RowSelfAttention
  // zeroes q at <pad> positions before the sum over alignment rows and fills the scores of the key columns that are <pad> in
  // ROW 0 with -10000 (finite; the softmax in the apply kernel sees ordinary numbers)
  using A = SplitAttn<MAXKB>;
  constexpr int tpad = A::tpad;
  __shared__ __attribute__((aligned(16))) char smem[2 * tpad * 128];
  char *Kh = smem, *Kl = smem + tpad * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kt = blockIdx.x % n_ktile, qc = (blockIdx.x / n_ktile) % n_qchunk, bh = blockIdx.x / (n_ktile * n_qchunk);
  const int b = bh / H, h = bh % H;
  const float* base = qkv + (size_t)b * R * C * ld_qkv + h * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int q0 = qc * 64 + wave * 16, k0 = kt * tpad;
  const bool active = q0 < C;
  const int qrow = q0 + fr < C ? q0 + fr : C - 1;
  f32x4 st[MAXKB];
#pragma unroll
  for (int kb = 0; kb < MAXKB; ++kb) st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < R; ++r) {
    const float* rb = base + (size_t)r * C * ld_qkv;
    bf16x8 qh[2], ql[2];
    A::load_q(rb + (size_t)qrow * ld_qkv, fq, qh, ql);
    if (tok && tok[((size_t)b * R + r) * C + qrow] == pad_idx) {
      const bf16x8 z = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
      qh[0] = z; qh[1] = z; ql[0] = z; ql[1] = z;
    }
    __syncthreads();
    A::stage(rb + (size_t)k0 * ld_qkv + k_off, ld_qkv, C - k0, Kh, Kl, tid);
    __syncthreads();
    if (active) A::qk(Kh, Kl, qh, ql, st, fr, fq);
  }
  const int q = q0 + fr;
  if (active && q < C) {
    float* dst = S + ((size_t)bh * C + q) * ldS + k0 + fq * 4;
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
      if (k0 + kb * 16 + fq * 4 < ldS) {
        float4 v = make_float4(st[kb][0] * scale, st[kb][1] * scale, st[kb][2] * scale, st[kb][3] * scale);
        if (tok) {
          const int32_t* t0 = tok + (size_t)b * R * C;            // row 0 of this MSA
          const int key = k0 + kb * 16 + fq * 4;
          if (key < C && t0[key] == pad_idx) v.x = -10000.f;
          if (key + 1 < C && t0[key + 1] == pad_idx) v.y = -10000.f;
          if (key + 2 < C && t0[key + 2] == pad_idx) v.z = -10000.f;
          if (key + 3 < C && t0[key + 3] == pad_idx) v.w = -10000.f;
        }
        *(float4*)(dst + kb * 16) = v;
      }
  }
}

// ---- tied row attention, step 2: ctx_r = softmax_j(S) v_r; one workgroup = (b, h, alignment row r, 64 queries) ---------
template <int MAXKB>
__global__ __launch_bounds__(256, 2) void msa_row_apply_split_kernel(const float* __restrict__ qkv, const float* __restrict__ S,
                                                                     bf16_t* __restrict__ ctx, int split_d, int R, int C, int H,
                                                                     int ld_qkv, int ld_ctx, int v_off, int n_qchunk, int ldS) {
  using A = SplitAttn<MAXKB>;
  constexpr int tpad = A::tpad;
  __shared__ __attribute__((aligned(16))) char smem[2 * tpad * 128];
  char *Vh = smem, *Vl = smem + tpad * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qc = blockIdx.x % n_qchunk, r = (blockIdx.x / n_qchunk) % R, bh = blockIdx.x / (n_qchunk * R);
  const int b = bh / H, h = bh % H;
  const float* rb = qkv + ((size_t)b * R + r) * C * ld_qkv + h * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int q0 = qc * 64 + wave * 16;
  const bool active = q0 < C;
  const int q = q0 + fr;
  const float* srow = S + ((size_t)bh * C + (q < C ? q : C - 1)) * ldS + fq * 4;
  f32x4 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;
  for (int k0 = 0; k0 < C; k0 += tpad) {
    f32x4 st[MAXKB];
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
      const int key0 = k0 + kb * 16 + fq * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key0 < ldS) v = *(const float4*)(srow + k0 + kb * 16);
      st[kb][0] = key0 < C ? v.x : -3.0e38f;
      st[kb][1] = key0 + 1 < C ? v.y : -3.0e38f;
      st[kb][2] = key0 + 2 < C ? v.z : -3.0e38f;
      st[kb][3] = key0 + 3 < C ? v.w : -3.0e38f;
    }
    __syncthreads();
    A::stage(rb + (size_t)k0 * ld_qkv + v_off, ld_qkv, C - k0, Vh, Vl, tid);
    __syncthreads();
    if (active) A::softmax_pv(st, o, m, l, Vh, Vl, fr, fq);
  }
  if (active && q < C) A::store_ctx(o, l, ctx + (((size_t)b * R + r) * C + q) * ld_ctx, h, fq, split_d);
}

__global__ __launch_bounds__(64) void attention_f32_kernel(const float* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                          int split_d, int T, int H, int ld_qkv_,
                                                          int ld_ctx_, int k_off, int v_off, SeqLayout sl, int n_qchunk,
                                                          const int32_t* __restrict__ key_tok, int pad_idx) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * 64];
  __shared__ __attribute__((aligned(16))) float Vs[64 * 64];
  const int lane = threadIdx.x;
  const int qc = blockIdx.x % n_qchunk;
  const int sh = blockIdx.x / n_qchunk;
  const int seq = sh / H, h = sh % H;
  const size_t row0 = (size_t)(seq / sl.inner_count) * sl.outer_rows + (size_t)(seq % sl.inner_count) * sl.inner_rows;
  const size_t ld_qkv = (size_t)ld_qkv_ * sl.row_step, ld_ctx = (size_t)ld_ctx_ * sl.row_step;
  const float* base = qkv + row0 * ld_qkv_ + h * 64;
  const int qi = qc * 64 + lane;
  const bool valid = qi < T;
  float q[64], o[64];
  {
    const float4* qp = (const float4*)(base + (size_t)(valid ? qi : T - 1) * ld_qkv);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 v = qp[i];
      q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 64; ++i) o[i] = 0.f;
  float m = -3.0e38f, l = 0.f;
  for (int k0 = 0; k0 < T; k0 += 64) {
    __syncthreads();
    const int nk = (T - k0) < 64 ? (T - k0) : 64;
    for (int i = lane; i < 64 * 16; i += 64) {       // 64 keys x 16 float4
      const int key = i >> 4, c = i & 15;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < nk) {
        kv = *(const float4*)(base + (size_t)(k0 + key) * ld_qkv + k_off + c * 4);
        vv = *(const float4*)(base + (size_t)(k0 + key) * ld_qkv + v_off + c * 4);
      }
      ((float4*)Ks)[i] = kv;
      ((float4*)Vs)[i] = vv;
    }
    __syncthreads();
    float s[64];
    float tmax = -3.0e38f;
#pragma unroll
    for (int key = 0; key < 64; ++key) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 kv = ((const float4*)Ks)[key * 16 + c];
        acc = fmaf(q[4 * c], kv.x, acc);
        acc = fmaf(q[4 * c + 1], kv.y, acc);
        acc = fmaf(q[4 * c + 2], kv.z, acc);
        acc = fmaf(q[4 * c + 3], kv.w, acc);
      }
      // keys beyond T, and <pad> keys of a ragged batch (key_tok = token buffer, sequence seq at seq*T)
      const bool masked = key >= nk || (key_tok && key_tok[(size_t)seq * T + k0 + key] == pad_idx);
      s[key] = masked ? -3.0e38f : acc;
      tmax = fmaxf(tmax, s[key]);
    }
    const float mn = fmaxf(m, tmax);
    const float alpha = expf(m - mn);
    l *= alpha;
#pragma unroll
    for (int i = 0; i < 64; ++i) o[i] *= alpha;
#pragma unroll
    for (int key = 0; key < 64; ++key) {
      const float p = s[key] > -1.0e38f ? expf(s[key] - mn) : 0.f;
      l += p;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 vv = ((const float4*)Vs)[key * 16 + c];
        o[4 * c] = fmaf(p, vv.x, o[4 * c]);
        o[4 * c + 1] = fmaf(p, vv.y, o[4 * c + 1]);
        o[4 * c + 2] = fmaf(p, vv.z, o[4 * c + 2]);
        o[4 * c + 3] = fmaf(p, vv.w, o[4 * c + 3]);
      }
    }
    m = mn;
  }
  if (valid) store_ctx64(o, l > 0.f ? 1.0f / l : 0.f, ctx + row0 * ld_ctx_ + (size_t)qi * ld_ctx, h, split_d);
}

// ---- strict tied row attention (MSA): S = scale * sum_r q_r k_r^T (fp32, to a scratch buffer), then
//      ctx[r] = softmax_j(S) v_r with the row statistics recomputed per thread ------------------------------
__global__ __launch_bounds__(64) void msa_row_scores_f32_kernel(const float* __restrict__ qkv, float* __restrict__ S, int R,
                                                               int C, int H, int ld_qkv, int k_off, float scale, int n_chunk, int ldS) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * 64];
  const int lane = threadIdx.x;
  const int jc = blockIdx.x % n_chunk, ic = (blockIdx.x / n_chunk) % n_chunk, bh = blockIdx.x / (n_chunk * n_chunk);
  const int b = bh / H, h = bh % H;
  const float* base = qkv + (size_t)b * R * C * ld_qkv + h * 64;
  const int qi = ic * 64 + lane;
  const int nk = (C - jc * 64) < 64 ? (C - jc * 64) : 64;
  float s[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) s[i] = 0.f;
  for (int r = 0; r < R; ++r) {
    const float* rb = base + (size_t)r * C * ld_qkv;
    __syncthreads();
    for (int i = lane; i < 64 * 16; i += 64) {
      const int key = i >> 4, c = i & 15;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key < nk) kv = *(const float4*)(rb + (size_t)(jc * 64 + key) * ld_qkv + k_off + c * 4);
      ((float4*)Ks)[i] = kv;
    }
    __syncthreads();
    float q[64];
    const float4* qp = (const float4*)(rb + (size_t)(qi < C ? qi : C - 1) * ld_qkv);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 v = qp[i];
      q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int key = 0; key < 64; ++key) {
      float acc = s[key];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 kv = ((const float4*)Ks)[key * 16 + c];
        acc = fmaf(q[4 * c], kv.x, acc);
        acc = fmaf(q[4 * c + 1], kv.y, acc);
        acc = fmaf(q[4 * c + 2], kv.z, acc);
        acc = fmaf(q[4 * c + 3], kv.w, acc);
      }
      s[key] = acc;
    }
  }
  if (qi < C) {
    float* dst = S + ((size_t)bh * C + qi) * ldS + jc * 64;
#pragma unroll
    for (int key = 0; key < 64; ++key)
      if (key < nk) dst[key] = s[key] * scale;
  }
}

__global__ __launch_bounds__(64) void msa_row_apply_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ S,
                                                              bf16_t* __restrict__ ctx, int split_d, int R,
                                                              int C, int H, int ld_qkv, int ld_ctx, int v_off, int n_chunk, int ldS) {
  __shared__ __attribute__((aligned(16))) float Vs[64 * 64];
  const int lane = threadIdx.x;
  const int ic = blockIdx.x % n_chunk, r = (blockIdx.x / n_chunk) % R, bh = blockIdx.x / (n_chunk * R);
  const int b = bh / H, h = bh % H;
  const float* rb = qkv + ((size_t)b * R + r) * C * ld_qkv + h * 64;
  const int qi = ic * 64 + lane;
  const bool valid = qi < C;
  const float* srow = S + ((size_t)bh * C + (valid ? qi : C - 1)) * ldS;
  float m = -3.0e38f;
  for (int j = 0; j < C; ++j) m = fmaxf(m, srow[j]);
  float l = 0.f;
  for (int j = 0; j < C; ++j) l += expf(srow[j] - m);
  float o[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) o[i] = 0.f;
  for (int k0 = 0; k0 < C; k0 += 64) {
    const int nk = (C - k0) < 64 ? (C - k0) : 64;
    __syncthreads();
    for (int i = lane; i < 64 * 16; i += 64) {
      const int key = i >> 4, c = i & 15;
      float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (key < nk) vv = *(const float4*)(rb + (size_t)(k0 + key) * ld_qkv + v_off + c * 4);
      ((float4*)Vs)[i] = vv;
    }
    __syncthreads();
    for (int key = 0; key < nk; ++key) {
      const float p = expf(srow[k0 + key] - m);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 vv = ((const float4*)Vs)[key * 16 + c];
        o[4 * c] = fmaf(p, vv.x, o[4 * c]);
        o[4 * c + 1] = fmaf(p, vv.y, o[4 * c + 1]);
        o[4 * c + 2] = fmaf(p, vv.z, o[4 * c + 2]);
        o[4 * c + 3] = fmaf(p, vv.w, o[4 * c + 3]);
      }
    }
  }
  if (valid) store_ctx64(o, l > 0.f ? 1.0f / l : 0.f, ctx + (((size_t)b * R + r) * C + qi) * ld_ctx, h, split_d);
}

static int attn_f32_mode() {
  static const int mode = [] { const char* e = getenv("PGIBBS_ATTN_F32"); return !e ? 0 : (e[0] == 'v' ? -1 : atoi(e)); }();
  return mode;
}

int launch_msa_row_attention_f32(hipStream_t s, const float* qkv, float* scores, bf16_t* ctx, int split_d, int B, int R,
                                 int C, int H, int ld_qkv, int ld_ctx, int k_off, int v_off, float scale, const int32_t* tok,
                                 int pad_idx) {
  if (B == 0 || R == 0) return 0;
  if (tok && attn_f32_mode() < 0) return fail(1, "row attention: the all-VALU cross-check kernels have no <pad> handling");
  const int n_chunk = (C + 63) / 64;
  const int ldS = msa_row_scores_ld(C);         // `scores` holds B*H*C rows of ldS floats
  if (attn_f32_mode() < 0) {
    hipLaunchKernelGGL(msa_row_scores_f32_kernel, dim3((unsigned)(B * H * n_chunk * n_chunk)), dim3(64), 0, s, qkv, scores, R, C, H,
                       ld_qkv, k_off, scale, n_chunk, ldS);
    hipLaunchKernelGGL(msa_row_apply_f32_kernel, dim3((unsigned)(B * H * R * n_chunk)), dim3(64), 0, s, qkv, scores, ctx, split_d,
                       R, C, H, ld_qkv, ld_ctx, v_off, n_chunk, ldS);
  } else {
    if (C <= 64) {
      hipLaunchKernelGGL(msa_row_scores_split_kernel<4>, dim3((unsigned)(B * H * n_chunk)), dim3(256), 0, s, qkv, scores, R, C, H,
                         ld_qkv, k_off, scale, n_chunk, 1, ldS, tok, pad_idx);
      hipLaunchKernelGGL(msa_row_apply_split_kernel<4>, dim3((unsigned)(B * H * R * n_chunk)), dim3(256), 0, s, qkv, scores, ctx,
                         split_d, R, C, H, ld_qkv, ld_ctx, v_off, n_chunk, ldS);
    } else {
      const int n_ktile = (C + 159) / 160;
      hipLaunchKernelGGL(msa_row_scores_split_kernel<10>, dim3((unsigned)(B * H * n_chunk * n_ktile)), dim3(256), 0, s, qkv, scores,
                         R, C, H, ld_qkv, k_off, scale, n_chunk, n_ktile, ldS, tok, pad_idx);
      hipLaunchKernelGGL(msa_row_apply_split_kernel<10>, dim3((unsigned)(B * H * R * n_chunk)), dim3(256), 0, s, qkv, scores, ctx,
                         split_d, R, C, H, ld_qkv, ld_ctx, v_off, n_chunk, ldS);
    }
  }
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_attention_f32(hipStream_t s, const float* qkv, bf16_t* ctx, int split_d, int64_t n_seq, int T, int H,
                         int ld_qkv, int ld_ctx, int k_off, int v_off, SeqLayout sl, const int32_t* key_tok, int pad_idx,
                         const float* bias_kv) {
  if (n_seq == 0) return 0;
  if (T <= 0) return fail(1, "attention: empty sequence");
  if (bias_kv && attn_f32_mode() < 0) return fail(1, "attention: the all-VALU cross-check kernel has no bias_k / bias_v key (ESM-1)");
  if (key_tok && sl.row_step != 1 && attn_f32_mode() < 0) return fail(1, "attention: the all-VALU cross-check kernel masks <pad> keys of contiguous chains only");
  const int mode = attn_f32_mode();
  // 16-query blocks per wave of the split kernel (its workgroup = 64 * nqb queries, all of them against each staged K / V tile):
  // the smallest of 1, 2, 3, 5 that covers the sequence with one workgroup, else 5 (PGIBBS_ATTN_F32_NQB overrides: A/B runs)
  static const int nqb_env = [] { const char* e = getenv("PGIBBS_ATTN_F32_NQB"); return e ? atoi(e) : 0; }();
  static const int kb_env = [] { const char* e = getenv("PGIBBS_ATTN_F32_KB"); return e ? atoi(e) : 0; }();   // experiment: key-tile height
  const int blocks = (T + 15) / 16;
  int nqb = blocks <= 4 ? 1 : (blocks <= 8 ? 2 : (blocks <= 12 ? 3 : 5));
  if (nqb_env == 1 || nqb_env == 2 || nqb_env == 3 || nqb_env == 5) nqb = nqb_env;
  if (mode < 0 || T < 64 || (T == 64 && !bias_kv)) nqb = 1;
  const int n_qchunk = (T + 64 * nqb - 1) / (64 * nqb);
  if (n_seq * H * n_qchunk > 0x7fffffff) return fail(1, "attention: too many sequences");
  const dim3 grid((unsigned)(n_seq * H * n_qchunk));
  if (mode < 0) {
    hipLaunchKernelGGL(attention_f32_kernel, grid, dim3(64), 0, s, qkv, ctx, split_d, T, H, ld_qkv, ld_ctx, k_off, v_off, sl,
                       n_qchunk, key_tok, pad_idx);
  } else {
    // key tile: 64 keys for short sequences, else 160 (80 KB of LDS: two workgroups per CU; at T = 258 a single 288-key
    // tile with one workgroup per CU was 1.6x slower)
    // Long plain sequences (five query blocks per wave): the tile height of 6, 8 or 10 key blocks that pads the keys least -- the
    // kernel's time follows the padded key count (T = 258: 3 x 96 = 288 keys 18.8 ms per config-2 iteration, 2 x 160 = 320 keys
    // 20.5, 3 x 128 = 384 keys 25.7); ties go to the taller tile.  PGIBBS_ATTN_F32_KB = 6 / 8 / 10 forces one.
    int kb5 = 10;
    {
      const int Tk = T + (bias_kv ? 1 : 0);            // every form: plain, <pad> mask, ESM-1's bias key (round 5)
      long best = ((long)Tk + 159) / 160 * 160;
      for (int k : {8, 6}) {
        const long padded = ((long)Tk + 16 * k - 1) / (16 * k) * (16 * k);
        if (padded < best) { best = padded; kb5 = k; }
      }
      if (kb_env == 6 || kb_env == 8 || kb_env == 10) kb5 = kb_env;
    }
#define PG_ATT_SPLIT(KB, NQ)                                                                                                  \
  do {                                                                                                                        \
    if (bias_kv && key_tok) hipLaunchKernelGGL((attention_split_kernel<KB, NQ, true, true>), grid, dim3(256), 0, s, qkv, ctx, split_d, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, \
                                               n_qchunk, key_tok, pad_idx, bias_kv);                                           \
    else if (bias_kv) hipLaunchKernelGGL((attention_split_kernel<KB, NQ, true, false>), grid, dim3(256), 0, s, qkv, ctx, split_d, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, \
                                         n_qchunk, key_tok, pad_idx, bias_kv);                                                 \
    else if (key_tok) hipLaunchKernelGGL((attention_split_kernel<KB, NQ, false, true>), grid, dim3(256), 0, s, qkv, ctx, split_d, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, \
                                         n_qchunk, key_tok, pad_idx, bias_kv);                                                 \
    else hipLaunchKernelGGL((attention_split_kernel<KB, NQ, false, false>), grid, dim3(256), 0, s, qkv, ctx, split_d, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, \
                            n_qchunk, key_tok, pad_idx, bias_kv);                                                              \
  } while (0)
    if (T < 64 || (T == 64 && !bias_kv)) PG_ATT_SPLIT(4, 1);
    else if (nqb == 1) PG_ATT_SPLIT(10, 1);
    else if (nqb == 2) PG_ATT_SPLIT(10, 2);
    else if (nqb == 3) PG_ATT_SPLIT(10, 3);
    else if (kb5 == 6) PG_ATT_SPLIT(6, 5);
    else if (kb5 == 8) PG_ATT_SPLIT(8, 5);
    else PG_ATT_SPLIT(10, 5);
#undef PG_ATT_SPLIT
  }
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
