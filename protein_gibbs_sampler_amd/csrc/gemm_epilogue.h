// Pieces shared by the GEMM tile kernels (gemm_bf16.hip, gemm_w16.hip; tools/probes/gemm_w4.hip): vector types, the GELU forms
// (one definition for all kernels) and the LDS-staged epilogue of the 256 x 256 tiles with 4 or 16 waves.
#pragma once
#include "kernels.h"

PG_OPS_BEGIN

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define PG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PG_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PG_NT_STORE(p, v) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)(p))

// ---- the GELU forms, defined ONCE for every tile kernel: a batch split into shards may pick different kernels per shard, and
// the logits must not depend on that ----
// erf-GELU, branch-free: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. below bf16/fp32 noise of the
// surrounding GEMM), one v_rcp + one v_exp instead of ocml's piecewise erff.  fp32 outputs (LM-head dense).
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float e = 1.0f - p * t * __expf(-z * z);       // erf(|x|/sqrt2)
  return 0.5f * x + 0.5f * fabsf(x) * e;                 // 0.5 x (1 + sign(x) erf(|x|/sqrt2))
}

// GELU of fc1 (bf16 outputs; strict mode's fused fc1 epilogue too): 0.5 x (1 + erf(x/sqrt2)) = relu(x) - |x| Phi(-|x|), with
// log2 Phi(-t) fitted by a degree-5 polynomial on t >= 0 (max abs error of the GELU 3.2e-6 -- three orders below the bf16
// rounding of the result -- and a small RELATIVE error in the negative tail; the leading coefficient is negative, so the power
// underflows to 0 for large t).  Two values at once: the polynomial and the final multiply-add as v_pk_fma_f32.  Per element
// 1 (abs) + 2.5 + 4 (quarter-rate v_exp_f32) + 0.5 + 1 (max) = 9 issue slots; the fc1 epilogue is VALU-bound.
typedef __attribute__((ext_vector_type(2))) float pg_f32x2;
#ifndef PG_STRICT_GELU_POLY
#define PG_STRICT_GELU_POLY 1
#endif
__device__ __forceinline__ pg_f32x2 gelu_poly2(float x0, float x1) {
  const pg_f32x2 t = {fabsf(x0), fabsf(x1)};
  pg_f32x2 p = {-4.074793151e-04f, -4.074793151e-04f};
  p = __builtin_elementwise_fma(p, t, (pg_f32x2){6.563348950e-03f, 6.563348950e-03f});
  p = __builtin_elementwise_fma(p, t, (pg_f32x2){-5.032995553e-02f, -5.032995553e-02f});
  p = __builtin_elementwise_fma(p, t, (pg_f32x2){-4.618885100e-01f, -4.618885100e-01f});
  p = __builtin_elementwise_fma(p, t, (pg_f32x2){-1.149779793e+00f, -1.149779793e+00f});
  p = __builtin_elementwise_fma(p, t, (pg_f32x2){-1.000206717e+00f, -1.000206717e+00f});
  const pg_f32x2 e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
  const pg_f32x2 r = {fmaxf(x0, 0.f), fmaxf(x1, 0.f)};
  return __builtin_elementwise_fma(-t, e, r);
}
__device__ __forceinline__ uint32_t gelu_bf16out_pack2(float x0, float x1) {
  const pg_f32x2 g = gelu_poly2(x0, x1);
  return pack_op2(g[0], g[1]);
}



template <int EPI> struct EpiTraits {
  static constexpr bool bf16out = EPI == EPI_BF16 || EPI == EPI_BF16_GELU;
  static constexpr bool gelu_bf16 = EPI == EPI_BF16_GELU;
  static constexpr bool resid = EPI == EPI_F32_RESID;
};

// ---------------------------------------------------------------------------------------------------------------------
// 64 x 64 "tail" tile computed by a whole 8- or 16-wave workgroup of the 256 x 256 kernels.  1290 tiles on 256 CUs are
// 5.04 rounds: the rows beyond the last full round used to go to a second, serial launch of the 64 x 64 kernel (3.5 ms of a
// 90 ms iteration at 100-330 TFLOP/s on an otherwise idle chip).  They are now extra workgroups at the FRONT of the same grid:
// one 64 x 64 tile each, K walked through a ring of eight 16-KB stages (the tile kernel's 128 KB of LDS) so that seven K-steps
// of LDS-DMA are in flight -- the tile is latency-, not throughput-bound -- and one s_barrier per K-step.  Same MFMA
// instruction and k order as every other tile kernel: a row's result does not depend on which tile shape computed it.
//   stage = 64 X rows + 64 W rows of 128 B = 16 pieces of 1 KiB (8 rows each); wave w stages piece w (and w + 8 with 8 waves)
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT3 (strict precision mode, gemm_w16.hip): operands in the split layout, K = logical depth; a K-step is one group of 32
// columns -- 128 B per row: xl | xh, wh | wl -- at a source stride of 192 B, and three products per step in the fused kernel's
// order (wh.xl, wl.xh, wh.xh).
template <int NW, int EPI, bool SPLIT3 = false>
__device__ __forceinline__ void gemm_tail_tile64(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                 const float* __restrict__ bias, void* __restrict__ out, int K, int ldx, int ldw,
                                                 int ldo, int m0, int n0, char* smem) {
  static_assert(NW == 8 || NW == 16, "tail tile: 8 or 16 waves");
  constexpr int STAGES = 8, STAGE_BYTES = 16384, PPW = 16 / NW, TM = 16 / NW;
  typedef EpiTraits<EPI> T;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = SPLIT3 ? K / 32 : K / 64;
  const int kbytes = (SPLIT3 ? 3 * K : K) * 2;           // bytes of an operand row
  constexpr int KSTEP_BYTES = SPLIT3 ? 192 : 128;

  // piece p (0-7: X rows 8p .., 8-15: W rows 8(p-8) ..): wave w stages piece w, and with 8 waves also piece w + 8 -- so piece A is
  // an X piece for w < 8 and piece B (8 waves only) always a W piece.  (No arrays of buffer resources: the type is opaque.)
  const bool a_is_w = wave >= 8;
  const int ld_a = a_is_w ? ldw : ldx;
  const bf16_t* src_a = a_is_w ? W + (size_t)(n0 + (wave - 8) * 8) * ldw : X + (size_t)(m0 + wave * 8) * ldx;
  const rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)src_a, 0, 7 * ld_a * 2 + kbytes, 0x00020000);
  const int voff_a = ((lane >> 3) * ld_a + ((lane & 7) ^ (lane >> 3)) * 8) * 2;
  const rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)(n0 + (wave & 7) * 8) * ldw), 0, 7 * ldw * 2 + kbytes, 0x00020000);
  const int voff_b = ((lane >> 3) * ldw + ((lane & 7) ^ (lane >> 3)) * 8) * 2;
  auto dma = [&](int t) {
    char* dst = smem + (t & (STAGES - 1)) * STAGE_BYTES;
    const int soff = t < nk ? t * KSTEP_BYTES : 0x7f000000;  // past the end of K: out of range, no memory traffic
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, PG_LDS_PTR(dst + wave * 1024), 16, voff_a, soff, 0, 0);
    if (PPW == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, PG_LDS_PTR(dst + (wave + 8) * 1024), 16, voff_b, soff, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int fo0 = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const int fo1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
  const int ni = NW == 16 ? (wave >> 2) : (wave >> 1);       // 16-row block of W (output columns)
  const int mi0 = NW == 16 ? (wave & 3) : (wave & 1) * 2;    // first 16-row block of X (output rows)
  f32x4 acc[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t) dma(t);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PPW) : "memory");     // my pieces of K-step t have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();          // everybody's have; everybody is done reading the slot of step t-1
    __builtin_amdgcn_sched_barrier(0);
    dma(t + STAGES - 1);                   // ... which the pieces of step t+7 now overwrite
    const char* sb = smem + (t & (STAGES - 1)) * STAGE_BYTES;
    if (SPLIT3) {
      const bf16x8 wh = *(const bf16x8*)(sb + 8192 + ni * 2048 + fo0), wl = *(const bf16x8*)(sb + 8192 + ni * 2048 + fo1);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const bf16x8 xl = *(const bf16x8*)(sb + (mi0 + j) * 2048 + fo0), xh = *(const bf16x8*)(sb + (mi0 + j) * 2048 + fo1);
        acc[j] = mfma_op16(wh, xl, acc[j]);
        acc[j] = mfma_op16(wl, xh, acc[j]);
        acc[j] = mfma_op16(wh, xh, acc[j]);
      }
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? fo1 : fo0;
      const bf16x8 wf = *(const bf16x8*)(sb + 8192 + ni * 2048 + fo);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const bf16x8 xf = *(const bf16x8*)(sb + (mi0 + j) * 2048 + fo);
        acc[j] = mfma_op16(wf, xf, acc[j]);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing zero-fill pieces, before the LDS is reused

  // lane holds D[n = ni*16 + fq*4 + r][m = (mi0 + j)*16 + fr], r = 0..3
  const int n_loc = ni * 16 + fq * 4;
  const float4 b4 = *(const float4*)(bias + n0 + n_loc);
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m_loc = (mi0 + j) * 16 + fr;
    float v0 = acc[j][0] + b4.x, v1 = acc[j][1] + b4.y, v2 = acc[j][2] + b4.z, v3 = acc[j][3] + b4.w;
    if (EPI == EPI_SPLIT3_GELU || EPI == EPI_SPLIT2_GELU) {
      // strict fc1: GELU, the value split into its bf16 (hi, lo) pair, written as fc2's operand row [lo | hi | hi] per 32 columns
      // (ldo = 3 N) -- the arithmetic of tile256_epilogue's EPI_SPLIT3_GELU branch
#if PG_STRICT_GELU_POLY
      const pg_f32x2 ga = gelu_poly2(v0, v1), gb = gelu_poly2(v2, v3);
      const float g0 = ga[0], g1 = ga[1], g2 = gb[0], g3 = gb[1];
#else
      const float g0 = gelu_erf(v0), g1 = gelu_erf(v1), g2 = gelu_erf(v2), g3 = gelu_erf(v3);
#endif
      uint2 hi, lo;
      hi.x = pack_op2(g0, g1);
      hi.y = pack_op2(g2, g3);
      lo.x = pack_op2(g0 - __uint_as_float(hi.x << 16), g1 - __uint_as_float(hi.x & 0xffff0000u));
      lo.y = pack_op2(g2 - __uint_as_float(hi.y << 16), g3 - __uint_as_float(hi.y & 0xffff0000u));
      const int n = n0 + n_loc;
      bf16_t* o3 = (bf16_t*)out + (size_t)(m0 + m_loc) * ldo + (n >> 5) * 96 + (n & 31);
      *(uint2*)o3 = lo;
      *(uint2*)(o3 + 32) = hi;
      if (EPI == EPI_SPLIT3_GELU) *(uint2*)(o3 + 64) = hi;
      continue;
    }
    const size_t o = (size_t)(m0 + m_loc) * ldo + n0 + n_loc;
    if (T::bf16out) {
      uint2 p;
      p.x = T::gelu_bf16 ? gelu_bf16out_pack2(v0, v1) : pack_op2(v0, v1);
      p.y = T::gelu_bf16 ? gelu_bf16out_pack2(v2, v3) : pack_op2(v2, v3);
      *(uint2*)((bf16_t*)out + o) = p;
    } else if (T::resid) {
      // x_old + (acc + bias), as the big tile's row-shaped epilogue adds them (fp32 addition commutes: same bits)
      float4* dst = (float4*)((float*)out + o);
      float4 r = *dst;
      r.x += v0; r.y += v1; r.z += v2; r.w += v3;
      *dst = r;
    } else {
      if (EPI == EPI_F32_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
      *(float4*)((float*)out + o) = make_float4(v0, v1, v2, v3);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of a 256 x 256 tile held by NW waves (4 or 16).  A lane holds 256 / NW "elements": four consecutive output features
// (n) of one token row (m).  elem(e, m_loc, n_loc) returns element e (compile-time after unrolling) and its position inside the
// tile.  The tile leaves through the (then idle) LDS so that every store instruction writes whole 512-B / 1-KiB output rows.
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI, int NW = 4, typename ElemF>
__device__ __forceinline__ void tile256_epilogue(ElemF&& elem, char* smem, int wave, int lane, int m0, int n0,
                                            const float* __restrict__ bias, void* __restrict__ out, int ldo) {
  constexpr int NE = 256 / NW;                   // elements per lane
  constexpr int RB = 256 / NW;                   // bf16 pass: tile rows owned by a wave
  constexpr int RF = 128 / NW;                   // fp32 passes: staged rows owned by a wave
  __syncthreads();                               // every wave is done with the operand ring
  if (EpiTraits<EPI>::bf16out) {
    // one pass: the whole 256 x 256 bf16 tile (128 KB) as 256 rows of 512 B, chunk-swizzled by row.  The GELU of fc1 is
    // applied in registers on the way in: +0.053 ms on the fc1 launch against +0.085 for fp32 staging in two passes with the
    // GELU on the way out (the ping-pong kernel's form); splitting the tile in two halves so that the stores of one drain
    // under the GELU of the other changed nothing (+0.059).
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int row, n;
      const f32x4 a = elem(e, row, n);
      const float4 b4 = *(const float4*)(bias + n0 + n);
      const float v0 = a[0] + b4.x, v1 = a[1] + b4.y, v2 = a[2] + b4.z, v3 = a[3] + b4.w;
      uint2 p;
      if (EpiTraits<EPI>::gelu_bf16) {
        p.x = gelu_bf16out_pack2(v0, v1);
        p.y = gelu_bf16out_pack2(v2, v3);
      } else {
        p.x = pack_op2(v0, v1);
        p.y = pack_op2(v2, v3);
      }
      *(uint2*)(smem + row * 512 + (((n >> 3) ^ (row & 31)) << 4) + (n & 4) * 2) = p;
    }
    __syncthreads();
    const int c = lane & 31;
    bf16_t* ob = (bf16_t*)out + (size_t)(m0 + wave * RB) * ldo + n0 + c * 8;
#pragma unroll
    for (int it = 0; it < RB / 2; ++it) {
      const int row = wave * RB + it * 2 + (lane >> 5);
      const uint4 v = *(const uint4*)(smem + row * 512 + ((c ^ (row & 31)) << 4));
      PG_NT_STORE((uint4*)(ob + (size_t)(it * 2 + (lane >> 5)) * ldo), v);
    }
    return;
  }
  if (EPI == EPI_SPLIT3_GELU || EPI == EPI_SPLIT2_GELU) {
    // Strict-mode fc1 (16-wave element order only): erf-GELU in registers, the value split into its bf16 (hi, lo) pair, and
    // the split operand rows of fc2 ([lo | hi | hi] per 32 columns, ldo = 3 N) written straight from here -- instead of an fp32 tile
    // plus a separate GELU-and-split pass over it (8 of 14 bytes per element less traffic).  Two halves of 128 token rows
    // (half h = tile rows with bit 5 == h = elements with bit 1 of e == h), each staged as a hi tile and a lo tile of 64 KB.
    static_assert((EPI != EPI_SPLIT3_GELU && EPI != EPI_SPLIT2_GELU) || NW == 16, "element order of the 16-wave kernel");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h) __syncthreads();
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        if (((e >> 1) & 1) != h) continue;
        int row, n;
        const f32x4 a = elem(e, row, n);
        const float4 b4 = *(const float4*)(bias + n0 + n);
#if PG_STRICT_GELU_POLY
        const pg_f32x2 ga = gelu_poly2(a[0] + b4.x, a[1] + b4.y), gb = gelu_poly2(a[2] + b4.z, a[3] + b4.w);
        const float g0 = ga[0], g1 = ga[1], g2 = gb[0], g3 = gb[1];
#else
        const float g0 = gelu_erf(a[0] + b4.x), g1 = gelu_erf(a[1] + b4.y), g2 = gelu_erf(a[2] + b4.z), g3 = gelu_erf(a[3] + b4.w);
#endif
        uint2 hi, lo;
        hi.x = pack_op2(g0, g1);
        hi.y = pack_op2(g2, g3);
        lo.x = pack_op2(g0 - __uint_as_float(hi.x << 16), g1 - __uint_as_float(hi.x & 0xffff0000u));
        lo.y = pack_op2(g2 - __uint_as_float(hi.y << 16), g3 - __uint_as_float(hi.y & 0xffff0000u));
        const int hr = (row >> 6) * 32 + (row & 31);
        const int off = hr * 512 + (((n >> 3) ^ (hr & 31)) << 4) + (n & 4) * 2;
        *(uint2*)(smem + off) = hi;
        *(uint2*)(smem + 65536 + off) = lo;
      }
      __syncthreads();
      const int c = lane & 31;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int hr = wave * 8 + it * 2 + (lane >> 5);
        const uint4 vh = *(const uint4*)(smem + hr * 512 + ((c ^ (hr & 31)) << 4));
        const uint4 vl = *(const uint4*)(smem + 65536 + hr * 512 + ((c ^ (hr & 31)) << 4));
        // columns n0 + 8c .. +7 of the token row -> group (n0 + 8c) / 32 of the split operand row, [lo | hi | hi] per 32 columns
        bf16_t* o = (bf16_t*)out + (size_t)(m0 + (hr >> 5) * 64 + h * 32 + (hr & 31)) * ldo + ((n0 >> 5) + (c >> 2)) * 96 + (c & 3) * 8;
        PG_NT_STORE((uint4*)o, vl);
        PG_NT_STORE((uint4*)(o + 32), vh);
        if (EPI == EPI_SPLIT3_GELU) PG_NT_STORE((uint4*)(o + 64), vh);      // EPI_SPLIT2_GELU: the duplicate block stays unwritten
      }
    }
    return;
  }
  // fp32 staging, two passes of 128 token rows x 1 KiB: pass p takes the elements whose token row has bit 6 == p (64 rows of
  // each 128-row half of the tile); wave w then owns the staged rows w*RF .. w*RF + RF-1 = token rows grow(p) .. grow(p) + RF-1
  auto stage = [&](int p) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      int row, n;
      const f32x4 a = elem(e, row, n);
      if (((row >> 6) & 1) != p) continue;       // row bit 6 is the same for all lanes of element e: wave-uniform, folds away
      const float4 b4 = *(const float4*)(bias + n0 + n);
      const int sr = (row >> 7) * 64 + (row & 63);
      float4 v = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
      if (EPI == EPI_F32_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      *(float4*)(smem + sr * 1024 + (((n >> 2) ^ (sr & 63)) << 4)) = v;
    }
  };
  // staged row sr <-> token row (sr >> 6) * 128 + p * 64 + (sr & 63); a wave's RF rows never straddle a 64-row block
  auto grow = [&](int p) { return m0 + ((wave * RF) >> 6) * 128 + p * 64 + ((wave * RF) & 63); };
  if (EPI == EPI_F32_RESID) {
    // out += tile: whole 1-KiB rows through buffer ops (wave-uniform row base, lane*16 offset); the row loads of the next
    // half of a wave's rows are in flight while the previous half is added and stored
    constexpr int HR = RF / 2;
    const int rstep = ldo * 4, voff = lane * 16;
    auto rs = [&](int p) { return __builtin_amdgcn_make_buffer_rsrc((float*)out + (size_t)grow(p) * ldo + n0, 0, 0x7fffffff, 0x00020000); };
    auto ldh = [&](f32x4 (&r)[HR], rsrc_t s, int first) {
#pragma unroll
      for (int it = 0; it < HR; ++it)
        r[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s, voff + (first + it) * rstep, 0, 2));
    };
    auto sth = [&](f32x4 (&r)[HR], rsrc_t s, int first) {
#pragma unroll
      for (int it = 0; it < HR; ++it) {
        const int sr = wave * RF + first + it;
        const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r[it] + v), s, voff + (first + it) * rstep, 0, 2);
      }
    };
    const rsrc_t rs0 = rs(0), rs1 = rs(1);
    f32x4 ra[HR], rb[HR];
    ldh(ra, rs0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage(0);
    __builtin_amdgcn_sched_barrier(0);
    ldh(rb, rs0, HR);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    sth(ra, rs0, 0);
    __builtin_amdgcn_sched_barrier(0);
    ldh(ra, rs1, 0);
    __builtin_amdgcn_sched_barrier(0);
    sth(rb, rs0, HR);
    __builtin_amdgcn_sched_barrier(0);
    ldh(rb, rs1, HR);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    stage(1);
    __syncthreads();
    sth(ra, rs1, 0);
    sth(rb, rs1, HR);
    return;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {                  // EPI_F32, EPI_F32_GELU
    if (p) __syncthreads();
    stage(p);
    __syncthreads();
    float* ob = (float*)out + (size_t)grow(p) * ldo + n0 + lane * 4;
#pragma unroll
    for (int it = 0; it < RF; ++it) {
      const int sr = wave * RF + it;
      const float4 v = *(const float4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
      PG_NT_STORE((float4*)(ob + (size_t)it * ldo), v);
    }
  }
}

PG_OPS_END
