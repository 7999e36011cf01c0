// Memory-bound kernels of the forward pass (HBM roofline): token embedding + token-dropout rescale +
// learned positions + LayerNorm, LayerNorm (fp32 residual -> bf16 GEMM operand), gather+LayerNorm of
// the sampled rows, and the LM-head tail (LayerNorm -> tied decoder -> logits).
//
// They restate, for the GPU, steps 1-4, the pre-LN of step 5, and steps 6-7 of the ESM-1b forward
// (fair-esm ProteinBertModel.forward) that the reference reaches through
// `self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_sampler.py:223), and step 1-2/4
// of MSATransformer.forward (A.3; esm_msa_sampler.py:136,236).
//
// Layout rule: one 64-lane wave per token row; a row of d fp32 is read as float4 per lane
// (16 B x 64 lanes = 1 KiB coalesced per instruction); mean/variance by wave shuffle reductions;
// nothing goes through LDS.
#include <stdlib.h>

#include "kernels.h"
#include "ln_row.h"

PG_OPS_BEGIN

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- embedding -------------------------------------------------------------------------------
// tokens[n_seq][T] -> x[n_seq*T][d] fp32 = LN_before(embed[tok]*scale + pos[...] (+ msa_row_pos[r]))
// token_dropout (ESM-1b): mask rows zeroed, all rows scaled by 0.88 / (1 - n_mask/src_len) per sequence.
// rows_per_msa > 0 (MSA-1b): adds msa_pos[seq % rows_per_msa].
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ embed,
                                                      const float* __restrict__ pos, const float* __restrict__ msa_pos,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ x, int64_t n_tok, int T, int d, int pad_idx,
                                                      int mask_idx, int token_dropout, int rows_per_msa, float eps,
                                                      const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                      bf16_t* __restrict__ h2, float embed_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_tok) return;
  const int64_t seq = row / T;
  const int t = (int)(row - seq * T);
  const int32_t* trow = tokens + seq * T;
  const int tok = trow[t];
  // per-sequence counts: masks, non-pad tokens, non-pad tokens at or before t
  int n_mask = 0, n_nonpad = 0, n_before = 0;
  for (int j = lane; j < T; j += 64) {
    const int tj = trow[j];
    n_mask += (tj == mask_idx);
    n_nonpad += (tj != pad_idx);
    n_before += (tj != pad_idx) && (j <= t);
  }
  n_mask = wave_sum_i(n_mask);
  n_nonpad = wave_sum_i(n_nonpad);
  n_before = wave_sum_i(n_before);
  const bool is_pad = (tok == pad_idx);
  float scale = embed_scale;               // 1 (ESM-1b, MSA-1b); sqrt(d) for ESM-1
  if (token_dropout) {
    scale = (1.0f - 0.15f * 0.8f) / (1.0f - (float)n_mask / (float)n_nonpad);
    if (tok == mask_idx) scale = 0.0f;
  }
  const int p = is_pad ? pad_idx : n_before + pad_idx;
  const int nch4 = d >> 2;
  const float4* e4 = (const float4*)(embed + (size_t)tok * d);
  const float4* p4 = (const float4*)(pos + (size_t)p * d);
  const float4* r4 = rows_per_msa > 0 ? (const float4*)(msa_pos + (size_t)(seq % rows_per_msa) * d) : nullptr;
  float4 v[kMaxCh];
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) {
      const int c = lane + 64 * i;
      const float4 e = e4[c], q = p4[c];
      v[i] = make_float4(e.x * scale + q.x, e.y * scale + q.y, e.z * scale + q.z, e.w * scale + q.w);
      if (r4) {
        const float4 r = r4[c];
        v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
      }
    }
  if (gamma) ln_inplace(v, nch4, lane, d, eps, gamma, beta);      // gamma == nullptr: no emb_layer_norm_before (ESM-1)
  float4* o = (float4*)(x + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) {
      if (is_pad) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      o[lane + 64 * i] = v[i];
    }
  if (h2) {      // the first layer's LayerNorm on the row just written (same values, same code as layernorm_bf16_kernel: same bits)
    ln_inplace(v, nch4, lane, d, eps, gamma2, beta2);
    store_row_bf16(h2 + (size_t)row * d, v, nch4, lane);
  }
}

// ---- LayerNorm: x fp32 [M][d] -> h bf16 [M][d] ------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ h,
                                                            int split3, int64_t M, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch4 = d >> 2;
  const float4* x4 = (const float4*)(x + (size_t)row * d);
  float4 v[kMaxCh];
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
  ln_inplace(v, nch4, lane, d, eps, gamma, beta);
  store_row_bf16(h + (size_t)row * d * (split3 ? 3 : 1), v, nch4, lane, split3, split3 != 2);       // split3 == 2: no duplicate hi block
}

// Mid-size batches (a 32 ... 128-chain shard of a multi-GPU job: 8 k ... 33 k token rows): with one row per wave the grid is 1.03 ...
// 4 "rounds" of what the chip holds at once (8 workgroups of 4 waves per CU), and the last, nearly empty round costs a whole
// row latency -- 14.9 us for 8448 rows where an eighth of the full-size launch is 10.4 (profiles/r05_shard_timeline_32chains.txt).
// Here the grid is exactly one resident round and every wave walks its rows (row, row + waves, ...) with the NEXT row's loads issued
// before the current row is normalised.  Same per-row arithmetic (ln_inplace, store_row_bf16): same bits.
// (the row width is a template parameter: with it the chunks a lane does not hold leave the register file -- 2 x 5 float4 at
// d = 1280 instead of 2 x 8 -- and eight workgroups per CU stay resident, as for the plain kernel)
template <int D>
__global__ __launch_bounds__(256) void layernorm_bf16_stride_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, bf16_t* __restrict__ h,
                                                                   int split3, int64_t M, float eps) {
  constexpr int d = D, nch4 = D >> 2;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4 v[kMaxCh], vn[kMaxCh];
  {
    const float4* x4 = (const float4*)(x + (size_t)row * d);
#pragma unroll
    for (int i = 0; i < kMaxCh; ++i)
      if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
  }
  for (;;) {
    const int64_t next = row + stride;
    if (next < M) {                                            // wave-uniform
      const float4* x4 = (const float4*)(x + (size_t)next * d);
#pragma unroll
      for (int i = 0; i < kMaxCh; ++i)
        if (lane + 64 * i < nch4) vn[i] = x4[lane + 64 * i];
    }
    ln_inplace(v, nch4, lane, d, eps, gamma, beta);
    store_row_bf16(h + (size_t)row * d * (split3 ? 3 : 1), v, nch4, lane, split3, split3 != 2);
    if (next >= M) break;
#pragma unroll
    for (int i = 0; i < kMaxCh; ++i) v[i] = vn[i];
    row = next;
  }
}

// The same with the output rows in COLUMN-MAJOR token order: token row (b*R + r)*C + c -> operand row (b*C + c)*R + r, so that a
// column's R tokens are contiguous rows for the fused column QKV + attention kernel (gemm_colattn.hip).  A separate kernel: the hot
// LayerNorm above stays as it is.
__global__ __launch_bounds__(256) void layernorm_bf16_colmajor_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, bf16_t* __restrict__ h,
                                                                     int64_t M, int d, float eps, int R, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch4 = d >> 2;
  const float4* x4 = (const float4*)(x + (size_t)row * d);
  float4 v[kMaxCh];
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
  ln_inplace(v, nch4, lane, d, eps, gamma, beta);
  const int64_t br = row / C;
  const int c = (int)(row - br * C);
  const int64_t b = br / R;
  const int r = (int)(br - b * R);
  store_row_bf16(h + (size_t)((b * C + c) * R + r) * d, v, nch4, lane);
}

// fp32 -> fp32 LayerNorm (debug entry / strict paths)
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y,
                                                           int64_t M, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch4 = d >> 2;
  const float4* x4 = (const float4*)(x + (size_t)row * d);
  float4 v[kMaxCh];
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
  ln_inplace(v, nch4, lane, d, eps, gamma, beta);
  float4* o = (float4*)(y + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) o[lane + 64 * i] = v[i];
}

// ---- gather the sampled rows + final LayerNorm -> bf16 (LM-head dense operand) ---------------
// sel r -> token row row_of(r) * width + idx[r]; idx < 0 (ragged padding) -> row of zeros.
// row_map == nullptr: selected row s = r / P maps to token row s.
__global__ __launch_bounds__(256) void gather_ln_bf16_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ row_map, int P, int width,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ h, int split3, int64_t n_sel, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_sel) return;
  const int nch4 = d >> 2;
  int pos = idx ? idx[r] : 0;
  float4 v[kMaxCh];
  if (pos < 0 || (idx && (pos & 0x3fffffff) >= width)) {
#pragma unroll
    for (int i = 0; i < kMaxCh; ++i) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    int64_t src;
    if (idx) {
      pos &= 0x3fffffff;
      const int64_t s = r / P;
      src = (row_map ? (int64_t)row_map[s] : s) * width + pos;
    } else {
      src = r;
    }
    const float4* x4 = (const float4*)(x + (size_t)src * d);
#pragma unroll
    for (int i = 0; i < kMaxCh; ++i)
      if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
    ln_inplace(v, nch4, lane, d, eps, gamma, beta);
  }
  store_row_bf16(h + (size_t)r * d * (split3 ? 3 : 1), v, nch4, lane, split3);
}

// ---- plain row gathers (last-layer pruning: only the sampled rows go through out-proj / FFN of the final layer) ----
// dst[r] = src[row_of(r) * width + idx[r]] for 16-byte chunks; idx < 0 -> row 0 (value never read back)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                         const int32_t* __restrict__ idx, const int32_t* __restrict__ row_map,
                                                         int P, int width, int64_t n_sel, int chunks,
                                                         const int32_t* __restrict__ d_iter) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_sel) return;
  if (d_iter) idx += (size_t)(*d_iter) * n_sel;
  int pos = idx[r];
  pos = pos < 0 ? 0 : (pos & 0x3fffffff);
  if (pos >= width) pos = 0;          // out-of-range entries are never sampled (sample_writeback_kernel skips them)
  const int64_t s = r / P;
  const int64_t row = (row_map ? (int64_t)row_map[s] : s) * width + pos;
  for (int c = threadIdx.x & 63; c < chunks; c += 64) dst[r * chunks + c] = src[row * chunks + c];
}

// ---- LM-head tail: logits[r][V] = LN(g[r]) . embed^T + bias ----------------------------------
// g = gelu(dense(x)) fp32 [n][d] (GEMM epilogue); the 33 x d tied decoder stays L2-resident.
__global__ __launch_bounds__(256) void lm_tail_kernel(const float* __restrict__ g, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ embed,
                                                     const float* __restrict__ out_bias, float* __restrict__ logits,
                                                     int64_t n, int d, int V, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int nch4 = d >> 2;
  const float4* x4 = (const float4*)(g + (size_t)r * d);
  float4 v[kMaxCh];
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
  if (gamma) ln_inplace(v, nch4, lane, d, eps, gamma, beta);      // gamma == nullptr: ESM-1's head has no LayerNorm
  float mine = 0.f;  // lane t keeps logit t (V <= 64)
  // four decoder rows per trip: their loads are all in flight before the first reduction (one row per trip was a chain
  // of V dependent L2 round trips: 81 us for a handful of rows)
  for (int t0 = 0; t0 < V; t0 += 4) {
    float s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u < V ? t0 + u : V - 1;
      const float4* e4 = (const float4*)(embed + (size_t)t * d);
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < kMaxCh; ++i)
        if (lane + 64 * i < nch4) {
          const float4 e = e4[lane + 64 * i];
          a += (v[i].x * e.x + v[i].y * e.y) + (v[i].z * e.z + v[i].w * e.w);
        }
      s[u] = a;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float r = wave_sum(s[u]);
      if (t0 + u < V && lane == t0 + u) mine = r + out_bias[t0 + u];
    }
  }
  if (lane < V) logits[(size_t)r * V + lane] = mine;
}

// Few rows (a single chain samples 2 positions per iteration; generate_single ~50): the row-per-wave kernel above walks the 33
// decoder rows one trip after the other (84 us for 2 rows).  Here one workgroup per row, one wave per 4 decoder rows: every wave
// normalises the row itself (1280 values) and all decoder rows are in flight at once.
__global__ __launch_bounds__(1024) void lm_tail_small_kernel(const float* __restrict__ g, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ embed,
                                                            const float* __restrict__ out_bias, float* __restrict__ logits,
                                                            int d, int V, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = blockIdx.x;
  const int nch4 = d >> 2;
  const float4* x4 = (const float4*)(g + (size_t)r * d);
  float4 v[kMaxCh];
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) v[i] = x4[lane + 64 * i];
  if (gamma) ln_inplace(v, nch4, lane, d, eps, gamma, beta);
  float s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int t = wave * 4 + u < V ? wave * 4 + u : V - 1;
    const float4* e4 = (const float4*)(embed + (size_t)t * d);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxCh; ++i)
      if (lane + 64 * i < nch4) {
        const float4 e = e4[lane + 64 * i];
        a += (v[i].x * e.x + v[i].y * e.y) + (v[i].z * e.z + v[i].w * e.w);      // same per-lane order as lm_tail_kernel
      }
    s[u] = a;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float tot = wave_sum(s[u]);
    const int t = wave * 4 + u;
    if (t < V && lane == 0) logits[(size_t)r * V + t] = tot + out_bias[t];
  }
}

// ---- strict mode helpers: fp32 -> (hi, lo) bf16 pair, optionally through erf-GELU ----------------
__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// src fp32 [rows][K] -> dst bf16 [rows][3K], per group of 32 columns [lo | hi | hi] (activation operand) or, WEIGHT, [hi | lo | hi]
// (layout: store_row_bf16)
template <bool GELU, bool WEIGHT>
__global__ void split3_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n4, int k4, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 v = ((const float4*)src)[i];
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    if (GELU) { v.x = gelu_erf_exact(v.x); v.y = gelu_erf_exact(v.y); v.z = gelu_erf_exact(v.z); v.w = gelu_erf_exact(v.w); }
    uint2 p, q;
    p.x = pack_op2(v.x, v.y);
    p.y = pack_op2(v.z, v.w);
    q.x = pack_op2(v.x - op16_to_f32((bf16_t)(p.x & 0xffff)), v.y - op16_to_f32((bf16_t)(p.x >> 16)));
    q.y = pack_op2(v.z - op16_to_f32((bf16_t)(p.y & 0xffff)), v.w - op16_to_f32((bf16_t)(p.y >> 16)));
    const int64_t row = i / k4;
    const int ci = (int)(i - row * k4);
    uint2* o = (uint2*)dst + row * 3 * k4 + (ci >> 3) * 24 + (ci & 7);
    o[0] = WEIGHT ? p : q;
    o[8] = WEIGHT ? q : p;
    o[16] = p;
  }
}
__global__ void gelu_f32_kernel(float* __restrict__ p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = gelu_erf_exact(p[i]);
}

// ---- fp32 <-> bf16 conversion (weights at load time, debug entries) ---------------------------
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = f32_to_op16_dev(src[i] * scale);
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = op16_to_f32(src[i]);
}
__global__ void scale_f32_kernel(float* __restrict__ p, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] *= scale;
}

// ---- launchers --------------------------------------------------------------------------------
static inline unsigned rows_grid(int64_t rows) { return (unsigned)((rows + 3) / 4); }

int launch_embed_ln(hipStream_t s, const int32_t* tokens, const float* embed, const float* pos, const float* msa_pos,
                    const float* gamma, const float* beta, float* x, int64_t n_tok, int T, int d, int pad_idx,
                    int mask_idx, int token_dropout, int rows_per_msa, float eps, const float* gamma2, const float* beta2,
                    bf16_t* h2, float embed_scale) {
  if (d % 4 || d > kMaxCh * 256) return fail(1, "embed: d must be a multiple of 4 and <= 2048");
  if (n_tok == 0) return 0;
  hipLaunchKernelGGL(embed_ln_kernel, dim3(rows_grid(n_tok)), dim3(256), 0, s, tokens, embed, pos, msa_pos, gamma, beta, x,
                     n_tok, T, d, pad_idx, mask_idx, token_dropout, rows_per_msa, eps, gamma2, beta2, h2, embed_scale);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_layernorm_bf16(hipStream_t s, const float* x, const float* gamma, const float* beta, bf16_t* h, int64_t M,
                          int d, float eps, bool split3, int colmajor_R, int colmajor_C, bool split3_dup) {
  if (d % 4 || d > kMaxCh * 256) return fail(1, "layernorm: d must be a multiple of 4 and <= 2048");
  if (M == 0) return 0;
  if (colmajor_R > 0) {
    if (split3 || M % ((int64_t)colmajor_R * colmajor_C)) return fail(1, "layernorm: column-major output needs whole MSAs and plain bf16 rows");
    hipLaunchKernelGGL(layernorm_bf16_colmajor_kernel, dim3(rows_grid(M)), dim3(256), 0, s, x, gamma, beta, h, M, d, eps, colmajor_R, colmajor_C);
    PG_HIP(hipGetLastError());
    return 0;
  }
  {
    // one resident round of workgroups walking their rows when the plain grid would be 1 ... 4 rounds (layernorm_bf16_stride_kernel)
    static const int on = [] { const char* e = getenv("PGIBBS_LN_STRIDE"); return e ? atoi(e) : 1; }();
    static const int n_cu = [] {
      hipDeviceProp_t p; int dv = 0; (void)hipGetDevice(&dv);
      return hipGetDeviceProperties(&p, dv) == hipSuccess ? p.multiProcessorCount : 256;
    }();
    const unsigned g1 = rows_grid(M);
    if (on && (d == 1280 || d == 768) && g1 > (unsigned)n_cu * 8 && g1 <= (unsigned)n_cu * 32) {
      // the grid = what is resident at once (occupancy of this very kernel x CUs, asked once per width)
      static int occ[2] = {0, 0};
      int& oc = occ[d == 768];
      if (oc == 0) {
        int v = 0;
        const hipError_t e = d == 1280 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, layernorm_bf16_stride_kernel<1280>, 256, 0)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, layernorm_bf16_stride_kernel<768>, 256, 0);
        oc = (e == hipSuccess && v > 0) ? v : -1;
      }
      if (oc > 0 && g1 > (unsigned)(n_cu * oc)) {
        const dim3 grid((unsigned)(n_cu * oc));
        const int sp = split3 ? (split3_dup ? 1 : 2) : 0;
        if (d == 1280) hipLaunchKernelGGL(layernorm_bf16_stride_kernel<1280>, grid, dim3(256), 0, s, x, gamma, beta, h, sp, M, eps);
        else hipLaunchKernelGGL(layernorm_bf16_stride_kernel<768>, grid, dim3(256), 0, s, x, gamma, beta, h, sp, M, eps);
        PG_HIP(hipGetLastError());
        return 0;
      }
    }
  }
  hipLaunchKernelGGL(layernorm_bf16_kernel, dim3(rows_grid(M)), dim3(256), 0, s, x, gamma, beta, h, split3 ? (split3_dup ? 1 : 2) : 0, M, d, eps);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_layernorm_f32(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int64_t M, int d,
                         float eps) {
  if (d % 4 || d > kMaxCh * 256) return fail(1, "layernorm: d must be a multiple of 4 and <= 2048");
  if (M == 0) return 0;
  hipLaunchKernelGGL(layernorm_f32_kernel, dim3(rows_grid(M)), dim3(256), 0, s, x, gamma, beta, y, M, d, eps);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_gather_ln_bf16(hipStream_t s, const float* x, const int32_t* idx, const int32_t* row_map, int P, int width,
                          const float* gamma, const float* beta, bf16_t* h, int64_t n_sel, int d, float eps, bool split3) {
  if (n_sel == 0) return 0;
  hipLaunchKernelGGL(gather_ln_bf16_kernel, dim3(rows_grid(n_sel)), dim3(256), 0, s, x, idx, row_map, P, width, gamma,
                     beta, h, split3 ? 1 : 0, n_sel, d, eps);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_gather_rows(hipStream_t s, const void* src, void* dst, const int32_t* idx, const int32_t* row_map, int P, int width,
                       int64_t n_sel, int row_bytes, const int32_t* d_iter) {
  if (n_sel == 0) return 0;
  if (row_bytes % 16) return fail(1, "gather: rows must be multiples of 16 bytes");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(n_sel)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, idx, row_map, P,
                     width, n_sel, row_bytes / 16, d_iter);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_lm_tail(hipStream_t s, const float* g, const float* gamma, const float* beta, const float* embed,
                   const float* out_bias, float* logits, int64_t n, int d, int V, float eps) {
  if (V > 64) return fail(1, "lm_tail: vocab > 64 unsupported");
  if (n == 0) return 0;
  // identical arithmetic per logit (same per-lane partial sums, same wave reduction): bit-equal results.  Up to 1024 rows (round 5;
  // it was 128): a 32-chain shard's 800 sampled rows took 81 us on the row-per-wave kernel -- 200 workgroups, each wave walking the
  // decoder rows in nine dependent trips -- PGIBBS_LM_TAIL_SMALL=n moves the switch
  static const int small_max = [] { const char* e = getenv("PGIBBS_LM_TAIL_SMALL"); return e ? atoi(e) : 1024; }();
  if (n <= small_max) {
    hipLaunchKernelGGL(lm_tail_small_kernel, dim3((unsigned)n), dim3(64 * ((V + 3) / 4)), 0, s, g, gamma, beta, embed, out_bias, logits,
                       d, V, eps);
    PG_HIP(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(lm_tail_kernel, dim3(rows_grid(n)), dim3(256), 0, s, g, gamma, beta, embed, out_bias, logits, n, d, V,
                     eps);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_split3_bf16(hipStream_t s, const float* src, bf16_t* dst, int64_t rows, int K, float scale, bool gelu, bool weight) {
  if (rows == 0) return 0;
  if (K % 32) return fail(1, "split: K must be a multiple of 32");
  const int64_t n4 = rows * (K / 4);
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  if (gelu && !weight) hipLaunchKernelGGL((split3_bf16_kernel<true, false>), dim3(grid), dim3(256), 0, s, src, dst, n4, K / 4, scale);
  else if (!gelu && weight) hipLaunchKernelGGL((split3_bf16_kernel<false, true>), dim3(grid), dim3(256), 0, s, src, dst, n4, K / 4, scale);
  else if (!gelu) hipLaunchKernelGGL((split3_bf16_kernel<false, false>), dim3(grid), dim3(256), 0, s, src, dst, n4, K / 4, scale);
  else return fail(1, "split: GELU on a weight operand");
  PG_HIP(hipGetLastError());
  return 0;
}
int launch_gelu_f32(hipStream_t s, float* p, int64_t n) {
  if (n == 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(gelu_f32_kernel, dim3(grid), dim3(256), 0, s, p, n);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_f32_to_bf16(hipStream_t s, const float* src, bf16_t* dst, int64_t n, float scale) {
  if (n == 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid), dim3(256), 0, s, src, dst, n, scale);
  PG_HIP(hipGetLastError());
  return 0;
}
int launch_bf16_to_f32(hipStream_t s, const bf16_t* src, float* dst, int64_t n) {
  if (n == 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid), dim3(256), 0, s, src, dst, n);
  PG_HIP(hipGetLastError());
  return 0;
}
int launch_scale_f32(hipStream_t s, float* p, int64_t n, float scale) {
  if (n == 0) return 0;
  const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(scale_f32_kernel, dim3(grid), dim3(256), 0, s, p, n, scale);
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
