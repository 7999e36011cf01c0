// One LayerNorm row on one 64-lane wave -- the ONLY definition of the arithmetic, shared by the stand-alone kernels
// (elementwise.hip) and by the residual GEMMs that normalise a finished row panel themselves (gemm_epilogue.h): a row's
// bf16 operand is bit-identical whichever kernel produced it.  Every floating-point operation is an explicitly rounded
// intrinsic, so no translation unit's contraction choices can differ.
#pragma once
#include "pg_common.h"

PG_OPS_BEGIN

constexpr int kMaxCh = 8;  // d <= 8 * 256 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = __fadd_rn(v, __shfl_xor(v, o));
  return v;
}

// normalise the row held in v[] (chunk c = lane + 64*i) in place: (x-mean)/sqrt(var+eps)*g + b
__device__ __forceinline__ void ln_inplace(float4 (&v)[kMaxCh], int nch4, int lane, int d, float eps,
                                           const float* __restrict__ gamma, const float* __restrict__ beta) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) s = __fadd_rn(s, __fadd_rn(__fadd_rn(v[i].x, v[i].y), __fadd_rn(v[i].z, v[i].w)));
  const float mean = __fdiv_rn(wave_sum(s), (float)d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) {
      v[i].x = __fsub_rn(v[i].x, mean); v[i].y = __fsub_rn(v[i].y, mean); v[i].z = __fsub_rn(v[i].z, mean); v[i].w = __fsub_rn(v[i].w, mean);
      q = __fadd_rn(q, __fadd_rn(__fmaf_rn(v[i].x, v[i].x, __fmul_rn(v[i].y, v[i].y)), __fmaf_rn(v[i].z, v[i].z, __fmul_rn(v[i].w, v[i].w))));
    }
  const float rstd = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(wave_sum(q), (float)d), eps)));
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) {
      const int c = lane + 64 * i;
      const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
      v[i].x = __fmaf_rn(__fmul_rn(v[i].x, rstd), g.x, b.x);
      v[i].y = __fmaf_rn(__fmul_rn(v[i].y, rstd), g.y, b.y);
      v[i].z = __fmaf_rn(__fmul_rn(v[i].z, rstd), g.z, b.z);
      v[i].w = __fmaf_rn(__fmul_rn(v[i].w, rstd), g.w, b.w);
    }
}

// hi = bf16(v).  split3 (strict precision mode): the row becomes the split-bf16 activation operand of 3 * d values, interleaved
// in groups of 32 columns: group g = [lo(32) | hi(32) | hi(32)] of columns 32g .. 32g+31, lo = bf16(v - hi).  Against a weight
// row packed [hi | lo | hi] the same way, one bf16 GEMM over K' = 3 d sums, per 32 columns and in this order,
// x_lo.w_hi + x_hi.w_lo + x_hi.w_hi in its fp32 accumulator -- whichever tile kernel runs it; the fused 16-wave kernel
// (gemm_w16.hip) reads only the first two blocks of each group and issues the same three products from registers.
// dup = false leaves the third block of every group unwritten (the consumer is the fused kernel: gemm_split3_fused).
__device__ __forceinline__ void store_row_bf16(bf16_t* dst, const float4 (&v)[kMaxCh], int nch4, int lane, bool split3 = false,
                                               bool dup = true) {
#pragma unroll
  for (int i = 0; i < kMaxCh; ++i)
    if (lane + 64 * i < nch4) {
      const int ci = lane + 64 * i;              // float4 index = columns 4 ci .. 4 ci + 3
      uint2 p;
      p.x = pack_op2(v[i].x, v[i].y);
      p.y = pack_op2(v[i].z, v[i].w);
      if (!split3) {
        ((uint2*)dst)[ci] = p;
      } else {
        uint2 q;
        q.x = pack_op2(__fsub_rn(v[i].x, op16_to_f32((bf16_t)(p.x & 0xffff))), __fsub_rn(v[i].y, op16_to_f32((bf16_t)(p.x >> 16))));
        q.y = pack_op2(__fsub_rn(v[i].z, op16_to_f32((bf16_t)(p.y & 0xffff))), __fsub_rn(v[i].w, op16_to_f32((bf16_t)(p.y >> 16))));
        uint2* o = (uint2*)dst + (ci >> 3) * 24 + (ci & 7);      // group of 32 columns = 96 values = 24 uint2
        o[0] = q;
        o[8] = p;
        if (dup) o[16] = p;
      }
    }
}

PG_OPS_END
