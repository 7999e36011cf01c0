// 256x256 bf16 MFMA GEMM tile for gfx950 with FOUR waves per workgroup (one per SIMD), each owning a 128 x 128 block of
// the tile in 256 accumulator registers:   out[M][N] (+)= X[M][K] . W[N][K]^T + bias[N]   (fp32 accumulate)
//
// Same op, operand layout and k order as gemm_bf16.hip (every dense layer behind `self.model.model(batch)`,
// /root/reference/src/pgen/esm_sampler.py:223); this is the large-batch kernel.  Why a second tile shape: the 8-wave kernel
// there gives a wave 128 x 64 outputs, i.e. 12 LDS fragment reads per 32 MFMAs, and its profile says the LDS side (DMA
// writes + fragment reads) outlasts the MFMA side.  A 128 x 128 wave tile needs 16 fragment reads per 64 MFMAs -- a third
// less LDS traffic per FLOP -- at the price of 256 accumulators + 128 fragment registers per lane, i.e. ONE wave per SIMD
// (the CU's whole 512-entry register file).  There is then no partner wave to hide LDS latency behind, so the loop is
// software-pipelined inside the wave: fragments of half-step s+1 are read into a second register set and the LDS-DMA pieces of
// half-step s+4 are issued between the 64 MFMAs of half-step s (about one non-MFMA instruction per three MFMAs).
//
//   * K is walked in half-steps of 32 (one v_mfma_f32_16x16x32_bf16 deep).  A half-step's operands are 256 X rows + 256 W
//     rows of 64 B = 32 KB of LDS = 32 DMA pieces of 1 KiB (16 rows each; a piece is exactly one MFMA fragment tile);
//     a ring of 4 such slots; the pieces of half-step s+4 are issued during half-step s (into the slot of s, whose
//     fragments are in registers by then) and waited for with a counted vmcnt three half-steps later.
//   * one s_barrier per half-step: it publishes the landed pieces of s+1 and retires everybody's reads of s-1's slot.
//   * 16-B chunks XOR-swizzled exactly as in gemm_bf16.hip (source-side permutation; conflict-free ds_read_b128).
//   * products of one output are accumulated in the same k order, by the same MFMA instruction, as in every other tile
//     kernel of this library -> results stay bit-identical however a batch is split over kernels, launches or GPUs.
//   * epilogues: through the (then idle) LDS ring so that every store instruction writes whole 512-B / 1-KiB output rows.
#include <stdlib.h>

#include "kernels.h"

namespace pg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define PG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PG_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define PG_NT_STORE(p, v) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)(p))

__device__ __forceinline__ float w4_gelu_erf(float x) {           // as gelu_erf in gemm_bf16.hip (fp32 outputs)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float e = 1.0f - p * t * __expf(-z * z);
  return 0.5f * x + 0.5f * fabsf(x) * e;
}
__device__ __forceinline__ float w4_gelu_bf16out(float x) {       // as gelu_bf16out in gemm_bf16.hip (bf16 outputs)
  const float t = fabsf(x);
  float p = -4.074793151e-04f;
  p = fmaf(p, t, 6.563348950e-03f);
  p = fmaf(p, t, -5.032995553e-02f);
  p = fmaf(p, t, -4.618885100e-01f);
  p = fmaf(p, t, -1.149779793e+00f);
  p = fmaf(p, t, -1.000206717e+00f);
  return fmaf(-t, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}

constexpr int W4_SLOT = 32 * 1024;     // one half-step: 256 X rows + 256 W rows of 64 B

struct W4Frags { bf16x8 w[8], x[8]; };

// The accumulate-in-place MFMA as an asm statement with a tied AGPR operand.  With the builtin and all 256 AGPRs holding
// accumulators, hipcc (ROCm 7.2) leaves srcC and vdst untied and "rotates" the tile through a[0:3]: four v_accvgpr_mov plus
// wait states in front of nearly every MFMA.  What the compiler then no longer knows (it sees an opaque statement): the
// MFMA -> non-MFMA-reader hazard of the accumulators, padded by hand (PG_W4_MFMA_DRAIN).
#define PG_W4_MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define PG_W4_MFMA_DRAIN() asm volatile("s_nop 11" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------------
// epilogue: acc[i][j] of wave (wm, wn) is D[n = n0 + wn*128 + i*16 + fq*4 + r][m = m0 + wm*128 + j*16 + fr]
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void w4_epilogue(f32x4 (&acc)[8][8], char* smem, int wm, int wn, int wave, int lane, int m0, int n0,
                                            const float* __restrict__ bias, void* __restrict__ out, int ldo) {
  const int fr = lane & 15, fq = lane >> 4;
  __syncthreads();                               // every wave is done with the operand ring
  if (EPI == EPI_BF16) {
    // one pass: the whole 256 x 256 bf16 tile (128 KB) as 256 rows of 512 B, chunk-swizzled by row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 b4 = *(const float4*)(bias + n0 + wn * 128 + i * 16 + fq * 4);
      const int c = wn * 16 + i * 2 + (fq >> 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = wm * 128 + j * 16 + fr;
        uint2 p;
        p.x = pack_bf16x2(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y);
        p.y = pack_bf16x2(acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
        *(uint2*)(smem + row * 512 + ((c ^ (row & 31)) << 4) + (fq & 1) * 8) = p;
      }
    }
    __syncthreads();
    const int c = lane & 31;
    bf16_t* ob = (bf16_t*)out + (size_t)(m0 + wave * 64) * ldo + n0 + c * 8;
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int row = wave * 64 + it * 2 + (lane >> 5);
      const uint4 v = *(const uint4*)(smem + row * 512 + ((c ^ (row & 31)) << 4));
      PG_NT_STORE((uint4*)(ob + (size_t)(it * 2 + (lane >> 5)) * ldo), v);
    }
    return;
  }
  // fp32 staging, two passes of 128 token rows x 1 KiB: pass p takes accumulator columns j = 4p..4p+3 of every wave;
  // wave w then owns the staged rows w*32 .. w*32+31 = token rows grow(p) .. grow(p)+31
  auto stage = [&](int p) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 b4 = *(const float4*)(bias + n0 + wn * 128 + i * 16 + fq * 4);
      const int c = wn * 32 + i * 4 + fq;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int sr = wm * 64 + jj * 16 + fr;
        const f32x4 a = acc[i][p * 4 + jj];
        float4 v = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
        if (EPI == EPI_F32_GELU) { v.x = w4_gelu_erf(v.x); v.y = w4_gelu_erf(v.y); v.z = w4_gelu_erf(v.z); v.w = w4_gelu_erf(v.w); }
        *(float4*)(smem + sr * 1024 + ((c ^ (sr & 63)) << 4)) = v;
      }
    }
  };
  auto grow = [&](int p) { return m0 + (wave >> 1) * 128 + p * 64 + (wave & 1) * 32; };
  if (EPI == EPI_BF16_GELU) {
    const int c8 = lane & 31;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p) __syncthreads();
      stage(p);
      __syncthreads();
      bf16_t* ob = (bf16_t*)out + (size_t)grow(p) * ldo + n0 + c8 * 8;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r2 = it * 2 + (lane >> 5), sr = wave * 32 + r2;
        const float4 a = *(const float4*)(smem + sr * 1024 + (((2 * c8) ^ (sr & 63)) << 4));
        const float4 b = *(const float4*)(smem + sr * 1024 + (((2 * c8 + 1) ^ (sr & 63)) << 4));
        uint4 v;
        v.x = pack_bf16x2(w4_gelu_bf16out(a.x), w4_gelu_bf16out(a.y));
        v.y = pack_bf16x2(w4_gelu_bf16out(a.z), w4_gelu_bf16out(a.w));
        v.z = pack_bf16x2(w4_gelu_bf16out(b.x), w4_gelu_bf16out(b.y));
        v.w = pack_bf16x2(w4_gelu_bf16out(b.z), w4_gelu_bf16out(b.w));
        PG_NT_STORE((uint4*)(ob + (size_t)r2 * ldo), v);
      }
    }
    return;
  }
  if (EPI == EPI_F32_RESID) {
    // out += tile: whole 1-KiB rows through buffer ops (wave-uniform row base, lane*16 offset); the row loads of the next
    // 16 rows are in flight while the previous 16 are added and stored
    const int rstep = ldo * 4, voff = lane * 16;
    auto rs = [&](int p) { return __builtin_amdgcn_make_buffer_rsrc((float*)out + (size_t)grow(p) * ldo + n0, 0, 0x7fffffff, 0x00020000); };
    auto ld16 = [&](f32x4 (&r)[16], rsrc_t s, int first) {
#pragma unroll
      for (int it = 0; it < 16; ++it)
        r[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s, voff + (first + it) * rstep, 0, 2));
    };
    auto st16 = [&](f32x4 (&r)[16], rsrc_t s, int first) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int sr = wave * 32 + first + it;
        const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r[it] + v), s, voff + (first + it) * rstep, 0, 2);
      }
    };
    const rsrc_t rs0 = rs(0), rs1 = rs(1);
    f32x4 ra[16], rb[16];
    ld16(ra, rs0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage(0);
    __builtin_amdgcn_sched_barrier(0);
    ld16(rb, rs0, 16);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    st16(ra, rs0, 0);
    __builtin_amdgcn_sched_barrier(0);
    ld16(ra, rs1, 0);
    __builtin_amdgcn_sched_barrier(0);
    st16(rb, rs0, 16);
    __builtin_amdgcn_sched_barrier(0);
    ld16(rb, rs1, 16);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    stage(1);
    __syncthreads();
    st16(ra, rs1, 0);
    st16(rb, rs1, 16);
    return;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {                  // EPI_F32, EPI_F32_GELU
    if (p) __syncthreads();
    stage(p);
    __syncthreads();
    float* ob = (float*)out + (size_t)grow(p) * ldo + n0 + lane * 4;
#pragma unroll
    for (int it = 0; it < 32; ++it) {
      const int sr = wave * 32 + it;
      const float4 v = *(const float4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
      PG_NT_STORE((float4*)(ob + (size_t)it * ldo), v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ABL (micro-benchmark ablations): 0 real kernel; 1 no LDS-DMA in the loop; 2 no MFMA; 3 no ds_read in the loop;
// 4 no epilogue stores (accumulators kept alive); 5 no barrier in the loop (timing only)
// ---------------------------------------------------------------------------------------------------------------------
// SCHED: MFMA order inside a half-step -- 0: W-fragment major (acc[g][0..7] for g = 0..7), 1: X-fragment-pair major
// DMA0: index of the MFMA pair behind which the first of the 8 DMA pieces is issued (16: right after the fragment reads)
// NS: ring slots (4 = 128 KB, 5 = all 160 KB of LDS); the DMA of half-step hs+NS is issued during half-step hs
template <int EPI, int GM, int ABL, int SCHED = 0, int DMA0 = 16, int NS = 4>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, void* __restrict__ out, int K,
                                                             int ldx, int ldw, int ldo, int tiles_n, int n_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[NS * W4_SLOT];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = n_tiles >> 3, r = n_tiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tile_m, tile_n;
  {
    const int tiles_m = n_tiles / tiles_n;
    const int gsz = GM * tiles_n, g = bid / gsz, within = bid - g * gsz;
    const int rows = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tile_m = g * GM + within % rows;
    tile_n = within / rows;
  }
  const int m0 = tile_m * 256, n0 = tile_n * 256;

  // ---- LDS-DMA: wave stages pieces wave*8 .. wave*8+7 of the 32 pieces of a half-step (0-15 X rows, 16-31 W rows).
  // Buffer form: one 32-bit lane offset for every piece; the piece / k offsets travel in the scalar offset.
  const bool stage_w = wave >= 2;
  const int ld_ = stage_w ? ldw : ldx;
  const bf16_t* src = (stage_w ? W + (size_t)n0 * ldw : X + (size_t)m0 * ldx) + (size_t)(wave & 1) * 128 * ld_;
  // num_records = this wave's 128-row band: half-steps past the end of K (the loop issues its prefetches unconditionally, see
  // below) get a scalar offset beyond it -> the hardware returns zeros without touching memory
  const int band_bytes = (127 * ld_ + K) * 2;
  const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, band_bytes, 0x00020000);
  const int schunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const int dma_voff = ((lane >> 2) * ld_ + schunk * 8) * 2;
  const int piece_bytes = 16 * ld_ * 2;
  const int lds_piece0 = wave * 8 * 1024;
  const int nh = K / 32;                           // half-steps

  auto dma_piece = [&](int hs, int slot_off, int g) {          // piece g (0..7) of this wave for half-step hs
    char* dst = smem + slot_off + lds_piece0 + g * 1024;
    const int soff = hs < nh ? g * piece_bytes + hs * 64 : 0x7f000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst), 16, dma_voff, soff, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);
  const int xoff = wm * 8 * 1024 + foff;
  const int woff = 16 * 1024 + wn * 8 * 1024 + foff;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int hs = 0; hs < NS; ++hs)
#pragma unroll
    for (int g = 0; g < 8; ++g) dma_piece(hs, hs * W4_SLOT, g);
  if (NS == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       // half-step 0 landed; 1 .. NS-1 stay in flight
  else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  W4Frags fa, fb;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    fa.w[g] = *(const bf16x8*)(smem + woff + g * 1024);
    fa.x[g] = *(const bf16x8*)(smem + xoff + g * 1024);
  }

  // One half-step: 64 MFMAs on `cur`; meanwhile read the fragments of half-step hs+1 into `nxt` and issue the DMA of hs+3.
  // Every half-step runs the SAME instruction stream -- also the last three, whose fragment reads fetch stale LDS that is
  // never used and whose DMA pieces are the no-traffic out-of-range loads above.  A peeled tail would be cheaper by three
  // 32 KB zero fills per tile, but hipcc assigns the 64 accumulator tuples differently there and connects the two
  // assignments with hundreds of v_accvgpr_mov placed right behind the (to it opaque) MFMA statements.
#define PG_W4_PHASE(HS, CUR, NXT)                                                                                          \
  {                                                                                                                        \
    /* pieces of hs+1 must have landed; those of hs+2 .. hs+NS-1 may stay in flight.  ABL 10: never wait (timing only) */  \
    if (ABL == 1 || ABL == 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
    else if (ABL == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
    else if (NS == 4) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");                                         \
    else asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    if (ABL != 5) __builtin_amdgcn_s_barrier();                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    /* after this barrier: slot `wr` (half-step hs, already in registers) is free for the DMA of hs+NS; slot `rd` holds */ \
    /* half-step hs+1 */                                                                                                   \
    const char* nb = smem + (ABL == 1 ? 0 : rd_off);                                                                       \
    /* 32 MFMA pairs; one other instruction behind each of the first 24: the 16 fragment reads of hs+1 (X tiles first: */ \
    /* the next half-step opens with w[0] against all of x[]), then this wave's 8 DMA pieces of hs+NS                   */ \
    _Pragma("unroll") for (int p = 0; p < 32; ++p) {                                                                       \
      const int g = SCHED == 0 ? p >> 2 : p & 7, j0 = SCHED == 0 ? (p & 3) * 2 : (p >> 3) * 2;                            \
      if (ABL != 2) {                                                                                                      \
        PG_W4_MFMA(acc[g][j0], CUR.w[g], CUR.x[j0]);                                                                       \
        PG_W4_MFMA(acc[g][j0 + 1], CUR.w[g], CUR.x[j0 + 1]);                                                               \
      } else if (p < 8) {                                                                                                  \
        asm volatile("" ::"v"(CUR.w[p]), "v"(CUR.x[p]));                                                                   \
      }                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      if (ABL != 3 && p < 8) NXT.x[p] = *(const bf16x8*)(nb + xoff + p * 1024);                                            \
      if (ABL != 3 && p >= 8 && p < 16) NXT.w[p - 8] = *(const bf16x8*)(nb + woff + (p - 8) * 1024);                      \
      if (ABL != 1 && p >= DMA0 && p < DMA0 + 8) dma_piece((HS) + NS, wr_off, p - DMA0);                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    if (ABL == 3) {                                                                                                        \
      _Pragma("unroll") for (int g = 0; g < 8; ++g) { NXT.w[g] = CUR.w[g]; NXT.x[g] = CUR.x[g]; }                          \
    }                                                                                                                      \
    wr_off = rd_off;                                                                                                       \
    rd_off = rd_off + W4_SLOT == NS * W4_SLOT ? 0 : rd_off + W4_SLOT;                                                      \
  }

  int wr_off = 0, rd_off = W4_SLOT;
  for (int hs = 0; hs < nh; hs += 2) {
    PG_W4_PHASE(hs, fa, fb)
    PG_W4_PHASE(hs + 1, fb, fa)
    // The compiler does not know that the asm statements above are MFMAs whose last results are still in the pipe, and its
    // register allocator places accumulator copies (v_accvgpr_*) on the loop-exit edge wherever it likes: pad the
    // MFMA -> VALU-read hazard (12 wait states for an 8-pass MFMA) INSIDE the loop body, 0.6 % of an iteration.
    PG_W4_MFMA_DRAIN();
  }
#undef PG_W4_PHASE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // retire the trailing (zero-fill) DMA pieces before LDS is reused
  __builtin_amdgcn_sched_barrier(0);

  if (ABL == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  w4_epilogue<EPI>(acc, smem, wm, wn, wave, lane, m0, n0, bias, out, ldo);
}

template <int ABL, int SCHED = 0, int DMA0 = 16, int NS = 4>
static int launch_w4_abl(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int K, int ldx, int ldw,
                         int ldo, int tiles_n, int n_tiles) {
  hipLaunchKernelGGL((gemm_bf16_w4_kernel<EPI_BF16, 4, ABL, SCHED, DMA0, NS>), dim3(n_tiles), dim3(256), 0, s, X, W, bias, out, K, ldx,
                     ldw, ldo, tiles_n, n_tiles);
  PG_HIP(hipGetLastError());
  return 0;
}

// M, N multiples of 256; K a multiple of 64, >= 128.  abl > 0: micro-benchmark variants (bf16 epilogue only)
int launch_gemm_w4(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldx,
                   int ldw, int ldo, int epi, int abl) {
  const int tiles_m = M / 256, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  if (M % 256 || N % 256 || K % 64 || K < 64 || n_tiles < 1) return fail(1, "gemm_w4: shape");
  switch (abl) {
    case 0: break;
    case 1: return launch_w4_abl<1>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 2: return launch_w4_abl<2>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 3: return launch_w4_abl<3>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 4: return launch_w4_abl<4>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 5: return launch_w4_abl<5>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 6: return launch_w4_abl<0, 1, 16>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);   // X-pair-major MFMA order
    case 7: return launch_w4_abl<0, 0, 24>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);   // DMA at the end of the half-step
    case 8: return launch_w4_abl<0, 0, 0>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);    // DMA first
    case 9: return launch_w4_abl<9>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);          // every DMA wait is vmcnt(0)
    case 10: return launch_w4_abl<10>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);        // DMA never waited for (timing)
    case 11: return launch_w4_abl<0, 0, 16, 5>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);   // 5-slot ring (160 KB)
    case 12: return launch_w4_abl<0, 0, 0, 5>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);    // 5 slots, DMA first
    default: return fail(1, "gemm_w4: unknown ablation");
  }
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  const int gm = gm_env ? gm_env : (K >= 4096 ? 2 : 4);
  dim3 grid(n_tiles), block(256);
#define PG_W4_CASE(E)                                                                                                      \
  case E:                                                                                                                  \
    if (gm == 2) hipLaunchKernelGGL((gemm_bf16_w4_kernel<E, 2, 0>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles); \
    else hipLaunchKernelGGL((gemm_bf16_w4_kernel<E, 4, 0>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);     \
    break;
  switch (epi) {
    PG_W4_CASE(EPI_BF16)
    PG_W4_CASE(EPI_BF16_GELU)
    PG_W4_CASE(EPI_F32_RESID)
    PG_W4_CASE(EPI_F32)
    PG_W4_CASE(EPI_F32_GELU)
    default:
      return fail(1, "gemm_w4: bad epilogue");
  }
#undef PG_W4_CASE
  PG_HIP(hipGetLastError());
  return 0;
}

}  // namespace pg
