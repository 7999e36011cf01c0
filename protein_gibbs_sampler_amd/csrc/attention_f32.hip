// Strict-precision attention (PG_PREC_FP32): the same softmax(q k^T) v per (sequence, head) as attention.hip, but
// every product and sum is an fp32 VALU op (no bf16 rounding of q, k, v or P), for the 1e-3 logit-parity mode.
// One 64-lane workgroup = 64 queries of one (sequence, head); a thread owns one query: q[64] and o[64] in
// registers, K/V tiles of 64 keys staged in LDS as fp32 (all lanes read the same K/V row -> LDS broadcast),
// online softmax per 64-key tile.  Throughput is irrelevant here (parity mode); correctness and accuracy are.
#include "kernels.h"

namespace pg {

// One thread's 64 context values of head h -> bf16 at dst (already offset to the head's columns).  split_d > 0: the row is
// the strict mode's K-concatenated operand [lo | hi | hi] (3 * split_d wide, elementwise.hip store_row_bf16).
__device__ __forceinline__ void store_ctx64(const float (&o)[64], float inv, bf16_t* dst, int split_d) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float a = o[4 * i] * inv, b = o[4 * i + 1] * inv, c = o[4 * i + 2] * inv, d = o[4 * i + 3] * inv;
    uint2 p, r;
    p.x = pack_bf16x2(a, b);
    p.y = pack_bf16x2(c, d);
    if (!split_d) {
      ((uint2*)dst)[i] = p;
    } else {
      r.x = pack_bf16x2(a - bf16_to_f32((bf16_t)(p.x & 0xffff)), b - bf16_to_f32((bf16_t)(p.x >> 16)));
      r.y = pack_bf16x2(c - bf16_to_f32((bf16_t)(p.y & 0xffff)), d - bf16_to_f32((bf16_t)(p.y >> 16)));
      ((uint2*)dst)[i] = r;
      ((uint2*)(dst + split_d))[i] = p;
      ((uint2*)(dst + 2 * split_d))[i] = p;
    }
  }
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
  if (valid) store_ctx64(o, 1.0f / l, ctx + row0 * ld_ctx_ + (size_t)qi * ld_ctx + h * 64, split_d);
}

// ---- strict tied row attention (MSA): S = scale * sum_r q_r k_r^T (fp32, to a scratch buffer), then
//      ctx[r] = softmax_j(S) v_r with the row statistics recomputed per thread ------------------------------
__global__ __launch_bounds__(64) void msa_row_scores_f32_kernel(const float* __restrict__ qkv, float* __restrict__ S, int R,
                                                               int C, int H, int ld_qkv, int k_off, float scale, int n_chunk) {
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
    float* dst = S + ((size_t)bh * C + qi) * C + jc * 64;
#pragma unroll
    for (int key = 0; key < 64; ++key)
      if (key < nk) dst[key] = s[key] * scale;
  }
}

__global__ __launch_bounds__(64) void msa_row_apply_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ S,
                                                              bf16_t* __restrict__ ctx, int split_d, int R,
                                                              int C, int H, int ld_qkv, int ld_ctx, int v_off, int n_chunk) {
  __shared__ __attribute__((aligned(16))) float Vs[64 * 64];
  const int lane = threadIdx.x;
  const int ic = blockIdx.x % n_chunk, r = (blockIdx.x / n_chunk) % R, bh = blockIdx.x / (n_chunk * R);
  const int b = bh / H, h = bh % H;
  const float* rb = qkv + ((size_t)b * R + r) * C * ld_qkv + h * 64;
  const int qi = ic * 64 + lane;
  const bool valid = qi < C;
  const float* srow = S + ((size_t)bh * C + (valid ? qi : C - 1)) * C;
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
  if (valid) store_ctx64(o, 1.0f / l, ctx + (((size_t)b * R + r) * C + qi) * ld_ctx + h * 64, split_d);
}

int launch_msa_row_attention_f32(hipStream_t s, const float* qkv, float* scores, bf16_t* ctx, int split_d, int B, int R,
                                 int C, int H, int ld_qkv, int ld_ctx, int k_off, int v_off, float scale) {
  if (B == 0 || R == 0) return 0;
  const int n_chunk = (C + 63) / 64;
  hipLaunchKernelGGL(msa_row_scores_f32_kernel, dim3((unsigned)(B * H * n_chunk * n_chunk)), dim3(64), 0, s, qkv, scores, R, C, H,
                     ld_qkv, k_off, scale, n_chunk);
  hipLaunchKernelGGL(msa_row_apply_f32_kernel, dim3((unsigned)(B * H * R * n_chunk)), dim3(64), 0, s, qkv, scores, ctx, split_d,
                     R, C, H, ld_qkv, ld_ctx, v_off, n_chunk);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_attention_f32(hipStream_t s, const float* qkv, bf16_t* ctx, int split_d, int64_t n_seq, int T, int H,
                         int ld_qkv, int ld_ctx, int k_off, int v_off, SeqLayout sl, const int32_t* key_tok, int pad_idx) {
  if (n_seq == 0) return 0;
  if (T <= 0) return fail(1, "attention: empty sequence");
  const int n_qchunk = (T + 63) / 64;
  if (n_seq * H * n_qchunk > 0x7fffffff) return fail(1, "attention: too many sequences");
  hipLaunchKernelGGL(attention_f32_kernel, dim3((unsigned)(n_seq * H * n_qchunk)), dim3(64), 0, s, qkv, ctx, split_d, T, H,
                     ld_qkv, ld_ctx, k_off, v_off, sl, n_qchunk, key_tok, pad_idx);
  PG_HIP(hipGetLastError());
  return 0;
}

}  // namespace pg
