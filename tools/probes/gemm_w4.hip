// PROBE, not part of libpgibbs.so (moved out of csrc/ in round 3: 5-10 % slower than the 8- / 16-wave kernels on every shape,
// DESIGN.md section 4 item 1).  To try it again: copy next to csrc/gemm_w16.hip, add it to the Makefile and declare launch_gemm_w4.
// 256x256 bf16 MFMA GEMM tile for gfx950 with FOUR waves per workgroup (one per SIMD), each owning a 128 x 128 block of
// the tile in 256 accumulator registers:   out[M][N] (+)= X[M][K] . W[N][K]^T + bias[N]   (fp32 accumulate)
//
// Same op, operand layout and k order as gemm_bf16.hip (every dense layer behind `self.model.model(batch)`,
// /root/reference/src/pgen/esm_sampler.py:223); this is the large-batch kernel.  Why a second tile shape: the 8-wave kernel
// there gives a wave 128 x 64 outputs, i.e. 12 LDS fragment reads per 32 MFMAs, and its profile says the LDS side (DMA
// writes + fragment reads) outlasts the MFMA side.  A 128 x 128 wave tile needs 16 fragment reads per 64 MFMAs -- a third
// less LDS traffic per FLOP -- at the price of 256 accumulators + 128 fragment registers per lane, i.e. ONE wave per SIMD
// (the CU's whole 512-entry register file).  There is then no partner wave to hide LDS latency behind, so the loop is
// software-pipelined inside the wave: fragments of the next half-step are read into a second register set and LDS-DMA pieces
// are issued between the MFMAs (about one non-MFMA instruction per three MFMAs).
//
//   * K is walked in K-steps of 64 = two half-steps of 32 (one v_mfma_f32_16x16x32_bf16 deep).  A K-step's operands are
//     256 X rows + 256 W rows of 128 B = 64 KB of LDS = 64 DMA pieces of 1 KiB (8 full 128-B rows each); two such slots.
//     The pieces of K-step t+2 are issued during the odd half-step of K-step t -- into t's own slot, whose fragments are all
//     in registers by then -- and waited for (vmcnt(0): nothing else is in flight) one K-step later.
//   * one s_barrier per K-step: it publishes the landed pieces of t+1 and retires everybody's reads of slot t.
//   * 16-B chunks of a 128-B row XOR-swizzled with (row & 7) (source-side permutation; conflict-free ds_read_b128).
//   * products of one output are accumulated in the same k order, by the same MFMA instruction, as in every other tile
//     kernel of this library -> results stay bit-identical however a batch is split over kernels, launches or GPUs.
//   * epilogues: through the (then idle) LDS ring so that every store instruction writes whole 512-B / 1-KiB output rows.
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace pg {

struct W4Frags { bf16x8 w[8], x[8]; };

// The accumulate-in-place MFMA as an asm statement with a tied AGPR operand.  With the builtin and all 256 AGPRs holding
// accumulators, hipcc (ROCm 7.2) leaves srcC and vdst untied and "rotates" the tile through a[0:3]: four v_accvgpr_mov plus
// wait states in front of nearly every MFMA.  What the compiler then no longer knows (it sees an opaque statement): the
// MFMA -> non-MFMA-reader hazard of the accumulators, padded by hand (PG_W4_MFMA_DRAIN).
#define PG_W4_MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define PG_W4_MFMA_DRAIN() asm volatile("s_nop 11" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------------
// ABL (micro-benchmark ablations): 0 real kernel; 1 no LDS-DMA in the loop; 2 no MFMA; 3 no ds_read in the loop;
// 4 no epilogue stores (accumulators kept alive); 5 no barrier in the loop (timing only); 9 (unused); 10 DMA never waited for
// SCHED: MFMA order inside a half-step -- 0: W-fragment major (acc[g][0..7] for g = 0..7), 1: X-fragment-pair major
// DMA0: index of the MFMA pair (0..31) of the ODD half-step behind which the first of the 16 DMA pieces is issued
// ---------------------------------------------------------------------------------------------------------------------
constexpr int W4_KSLOT = 64 * 1024;    // one K-step (64 deep): 256 X rows + 256 W rows of 128 B

template <int EPI, int GM, int ABL, int SCHED = 0, int DMA0 = 0>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, void* __restrict__ out, int K,
                                                             int ldx, int ldw, int ldo, int tiles_n, int n_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[2 * W4_KSLOT];

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

  // ---- LDS-DMA.  A piece (one wave-instruction, 64 lanes x 16 B = 1 KiB) is 8 operand rows x 128 B: every 128-B line of X / W
  // is fetched whole by ONE instruction (64-B half rows -- a piece per half-step -- pull every line through the L2 -> L1 path
  // twice; measured: the DMA stream then caps at ~10 TB/s chip-wide and its issue stalls cost the one-wave-per-SIMD loop 30 %).
  // A K-step slot holds 64 pieces (0-31 X rows, 32-63 W rows); wave w stages pieces 16w .. 16w+15, i.e. its own 128-row band.
  // Buffer form: one 32-bit lane offset for every piece, the piece / k offsets travel in the scalar offset; num_records = the
  // band, so K-steps past the end of K (the loop prefetches unconditionally) read zeros without touching memory.
  const bool stage_w = wave >= 2;
  const int ld_ = stage_w ? ldw : ldx;
  const bf16_t* src = (stage_w ? W + (size_t)n0 * ldw : X + (size_t)m0 * ldx) + (size_t)(wave & 1) * 128 * ld_;
  const int band_bytes = (127 * ld_ + K) * 2;
  const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, band_bytes, 0x00020000);
  const int schunk = (lane & 7) ^ (lane >> 3);               // 16-B chunks of a row XOR-swizzled by (row & 7)
  const int dma_voff = ((lane >> 3) * ld_ + schunk * 8) * 2;
  const int piece_bytes = 8 * ld_ * 2;
  const int lds_piece0 = wave * 16 * 1024;
  const int nk = K / 64;                                     // K-steps

  auto dma_piece = [&](int t, int g) {                       // piece g (0..15) of this wave for K-step t
    char* dst = smem + (t & 1) * W4_KSLOT + lds_piece0 + g * 1024;
    const int soff = t < nk ? g * piece_bytes + t * 128 : 0x7f000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst), 16, dma_voff, soff, 0, 0);
  };

  // fragment tile T (16 rows x 128 B = 2 KiB) of half kk: lane reads row fr, chunk (kk*4 + fq) ^ (fr & 7)
  const int fr = lane & 15, fq = lane >> 4;
  const int foff0 = fr * 128 + ((fq ^ (fr & 7)) << 4);       // kk = 0
  const int foff1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4); // kk = 1
  const int xbase = wm * 8 * 2048;
  const int wbase = 32 * 1024 + wn * 8 * 2048;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) dma_piece(t, g);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // K-step 0 landed; K-step 1 stays in flight
  __builtin_amdgcn_s_barrier();

  W4Frags fa, fb;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    fa.w[g] = *(const bf16x8*)(smem + wbase + g * 2048 + foff0);
    fa.x[g] = *(const bf16x8*)(smem + xbase + g * 2048 + foff0);
  }

  // One half-step = 64 MFMAs on CUR, one other instruction behind each of the first MFMA pairs:
  //   even half-step (t, 0): the 16 fragment reads of (t, 1), same slot;
  //   odd  half-step (t, 1): top: the pieces of K-step t+1 (issued one K-step ago) have landed -> barrier -> the 16 fragment
  //                          reads of (t+1, 0) from the other slot, and the 16 DMA pieces of K-step t+2 into THIS slot, whose
  //                          last fragment reads every wave has retired before that barrier.
  // Every K-step runs the same instruction stream, also the last two (their reads fetch stale LDS that is never used, their
  // DMA pieces are the out-of-range zero fills): a peeled tail makes hipcc re-assign the 64 accumulator tuples and connect
  // the two assignments with hundreds of v_accvgpr_mov right behind the -- to it opaque -- MFMA statements.
#define PG_W4_HALF(ODD, T, CUR, NXT)                                                                                       \
  {                                                                                                                        \
    if ((ODD) && ABL != 1 && ABL != 10) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    if (ABL != 5 && (ODD)) __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    const char* nb = smem + (((ODD) && ABL != 1 ? (T) + 1 : (T)) & 1) * W4_KSLOT;                                          \
    const int nfoff = (ODD) ? foff0 : foff1;                                                                               \
    _Pragma("unroll") for (int p = 0; p < 32; ++p) {                                                                       \
      const int g = SCHED != 1 ? p >> 2 : p & 7, j0 = SCHED != 1 ? (p & 3) * 2 : (p >> 3) * 2;                            \
      if (ABL != 2) {                                                                                                      \
        PG_W4_MFMA(acc[g][j0], CUR.w[g], CUR.x[j0]);                                                                       \
        if (SCHED == 2 && (ODD) && ABL != 1) {   /* wave-staggered DMA slots: MFMA 2p belongs to wave (2p & 3) */           \
          __builtin_amdgcn_sched_barrier(0);                                                                               \
          if (wave == ((2 * p) & 3)) dma_piece((T) + 2, p >> 1);                                                           \
          __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                  \
        PG_W4_MFMA(acc[g][j0 + 1], CUR.w[g], CUR.x[j0 + 1]);                                                               \
        if (SCHED == 2 && (ODD) && ABL != 1) {                                                                             \
          __builtin_amdgcn_sched_barrier(0);                                                                               \
          if (wave == ((2 * p + 1) & 3)) dma_piece((T) + 2, p >> 1);                                                       \
          __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                  \
      } else if (p < 8) {                                                                                                  \
        asm volatile("" ::"v"(CUR.w[p]), "v"(CUR.x[p]));                                                                   \
      }                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      if (ABL != 3 && p < 8) NXT.x[p] = *(const bf16x8*)(nb + xbase + p * 2048 + nfoff);                                   \
      if (ABL != 3 && p >= 8 && p < 16) NXT.w[p - 8] = *(const bf16x8*)(nb + wbase + (p - 8) * 2048 + nfoff);             \
      if (SCHED != 2 && (ODD) && ABL != 1 && p >= DMA0 && p < DMA0 + 16) dma_piece((T) + 2, p - DMA0);                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    if (ABL == 3) {                                                                                                        \
      _Pragma("unroll") for (int g = 0; g < 8; ++g) { NXT.w[g] = CUR.w[g]; NXT.x[g] = CUR.x[g]; }                          \
    }                                                                                                                      \
  }

  for (int t = 0; t < nk; ++t) {
    PG_W4_HALF(false, t, fa, fb)
    PG_W4_HALF(true, t, fb, fa)
    // The compiler does not know that the asm statements above are MFMAs whose last results are still in the pipe, and its
    // register allocator places accumulator copies (v_accvgpr_*) on the loop-exit edge wherever it likes: pad the
    // MFMA -> VALU-read hazard (12 wait states for an 8-pass MFMA) INSIDE the loop body, 0.6 % of an iteration.
    PG_W4_MFMA_DRAIN();
  }
#undef PG_W4_HALF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // retire the trailing (zero-fill) DMA pieces before LDS is reused
  __builtin_amdgcn_sched_barrier(0);

  if (ABL == 4) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // acc[i][j] is D[n = wn*128 + i*16 + fq*4 + r][m = wm*128 + j*16 + fr]
  auto elem = [&](int e, int& m_loc, int& n_loc) -> f32x4 {
    m_loc = wm * 128 + (e & 7) * 16 + fr;
    n_loc = wn * 128 + (e >> 3) * 16 + fq * 4;
    return acc[e >> 3][e & 7];
  };
  tile256_epilogue<EPI, 4>(elem, smem, wave, lane, m0, n0, bias, out, ldo);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tile with v_mfma_f32_32x32x16_bf16 (a wave's 128 x 128 block = 4 x 4 accumulators of 16 registers).  One MFMA
// occupies the matrix pipe for 32 cycles instead of 16, so the issue stall of an LDS-DMA instruction (~60 cycles when the
// CU's four waves issue theirs in lockstep) hides behind the two MFMAs in flight instead of draining the pipe.
// Operand rows are XOR-swizzled with (row >> 1) & 7 here (fragments are 32 rows x 16 k: lanes 0-31 read 32 rows).
// NOTE: a 32x32x16 MFMA adds its 16 products in a different order than the 16x16x32 one: results are NOT bit-identical to
// the other tile kernels' (same k order between MFMAs, different order inside).
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
struct W4Frags32 { bf16x8 w[2][4], x[2][4]; };
#define PG_W4_MFMA32(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))

template <int EPI, int GM, int ABL, int DMA0 = 16>
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4m32_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                const float* __restrict__ bias, void* __restrict__ out, int K,
                                                                int ldx, int ldw, int ldo, int tiles_n, int n_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[2 * W4_KSLOT];

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

  // LDS-DMA as in gemm_bf16_w4_kernel; a piece is rows 8g .. 8g+7 of the wave's band, i.e. rows (g & 3) * 8 + r of a 32-row
  // block: swizzle key (row >> 1) & 7 = (g & 1) * 4 + (r >> 1) -> one lane offset for even pieces, one for odd pieces
  const bool stage_w = wave >= 2;
  const int ld_ = stage_w ? ldw : ldx;
  const bf16_t* src = (stage_w ? W + (size_t)n0 * ldw : X + (size_t)m0 * ldx) + (size_t)(wave & 1) * 128 * ld_;
  const int band_bytes = (127 * ld_ + K) * 2;
  const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, band_bytes, 0x00020000);
  const int r8 = lane >> 3;
  const int dma_voff_e = (r8 * ld_ + ((lane & 7) ^ (r8 >> 1)) * 8) * 2;
  const int dma_voff_o = (r8 * ld_ + ((lane & 7) ^ (4 + (r8 >> 1))) * 8) * 2;
  const int piece_bytes = 8 * ld_ * 2;
  const int lds_piece0 = wave * 16 * 1024;
  const int nk = K / 64;

  auto dma_piece = [&](int t, int g) {
    char* dst = smem + (t & 1) * W4_KSLOT + lds_piece0 + g * 1024;
    const int soff = t < nk ? g * piece_bytes + t * 128 : 0x7f000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst), 16, (g & 1) ? dma_voff_o : dma_voff_e, soff, 0, 0);
  };

  // fragment of block B (32 rows x 128 B = 4 KiB), half kk, k16-step s2: lane reads row (lane & 31), chunk (kk*4 + s2*2 + (lane >> 5)) ^ key
  const int fr = lane & 31, fh = lane >> 5, key = (fr >> 1) & 7;
  int foff[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) foff[kk][s2] = fr * 128 + (((kk * 4 + s2 * 2 + fh) ^ key) << 4);
  const int xbase = wm * 4 * 4096;
  const int wbase = 32 * 1024 + wn * 4 * 4096;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[i][j][c] = 0.f;

#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) dma_piece(t, g);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  W4Frags32 fa, fb;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      fa.x[s2][b] = *(const bf16x8*)(smem + xbase + b * 4096 + foff[0][s2]);
      fa.w[s2][b] = *(const bf16x8*)(smem + wbase + b * 4096 + foff[0][s2]);
    }

  // half-step: 32 MFMAs (s2 = 0: 16, s2 = 1: 16); behind MFMA q < 16 the q-th fragment read of the next half-step (order
  // x[0][0..3], w[0][0..3], x[1][..], w[1][..]); in odd half-steps behind MFMA DMA0 + g the DMA piece g of K-step t + 2
#define PG_W4_HALF32(ODD, T, CUR, NXT)                                                                                     \
  {                                                                                                                        \
    if ((ODD) && ABL != 1 && ABL != 10) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    if (ABL != 5 && (ODD)) __builtin_amdgcn_s_barrier();                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    const char* nb = smem + (((ODD) && ABL != 1 ? (T) + 1 : (T)) & 1) * W4_KSLOT;                                          \
    _Pragma("unroll") for (int q = 0; q < 32; ++q) {                                                                       \
      const int s2 = q >> 4, bi = (q >> 2) & 3, bj = q & 3;                                                                \
      if (ABL != 2) PG_W4_MFMA32(acc[bi][bj], CUR.w[s2][bi], CUR.x[s2][bj]);                                               \
      else if (q < 8) asm volatile("" ::"v"(CUR.w[q >> 2][q & 3]), "v"(CUR.x[q >> 2][q & 3]));                             \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      if (ABL != 3 && q < 16) {                                                                                            \
        const int ns2 = q >> 3, nbk = q & 3;                                                                               \
        if (q & 4) NXT.w[ns2][nbk] = *(const bf16x8*)(nb + wbase + nbk * 4096 + foff[(ODD) ? 0 : 1][ns2]);                 \
        else NXT.x[ns2][nbk] = *(const bf16x8*)(nb + xbase + nbk * 4096 + foff[(ODD) ? 0 : 1][ns2]);                       \
      }                                                                                                                    \
      if ((ODD) && ABL != 1 && q >= DMA0 && q < DMA0 + 16) dma_piece((T) + 2, q - DMA0);                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }                                                                                                                      \
    if (ABL == 3) {                                                                                                        \
      _Pragma("unroll") for (int g = 0; g < 8; ++g) { NXT.w[g >> 2][g & 3] = CUR.w[g >> 2][g & 3]; NXT.x[g >> 2][g & 3] = CUR.x[g >> 2][g & 3]; } \
    }                                                                                                                      \
  }

  for (int t = 0; t < nk; ++t) {
    PG_W4_HALF32(false, t, fa, fb)
    PG_W4_HALF32(true, t, fb, fa)
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // MFMA -> VALU-read hazard, see gemm_bf16_w4_kernel
  }
#undef PG_W4_HALF32
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  if (ABL == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // element e = (bi*4 + bj)*4 + q: components 4q .. 4q+3 of acc[bi][bj] = D[n = wn*128 + bi*32 + q*8 + fh*4 + r][m = wm*128 + bj*32 + fr]
  auto elem = [&](int e, int& m_loc, int& n_loc) -> f32x4 {
    const int q = e & 3, bj = (e >> 2) & 3, bi = e >> 4;
    m_loc = wm * 128 + bj * 32 + fr;
    n_loc = wn * 128 + bi * 32 + q * 8 + fh * 4;
    const f32x16 a = acc[bi][bj];
    return (f32x4){a[q * 4], a[q * 4 + 1], a[q * 4 + 2], a[q * 4 + 3]};
  };
  tile256_epilogue<EPI, 4>(elem, smem, wave, lane, m0, n0, bias, out, ldo);
}

template <int ABL, int DMA0 = 16>
static int launch_w4m32_abl(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int K, int ldx, int ldw,
                            int ldo, int tiles_n, int n_tiles) {
  hipLaunchKernelGGL((gemm_bf16_w4m32_kernel<EPI_BF16, 4, ABL, DMA0>), dim3(n_tiles), dim3(256), 0, s, X, W, bias, out, K, ldx, ldw,
                     ldo, tiles_n, n_tiles);
  PG_HIP(hipGetLastError());
  return 0;
}

template <int ABL, int SCHED = 0, int DMA0 = 0>
static int launch_w4_abl(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int K, int ldx, int ldw,
                         int ldo, int tiles_n, int n_tiles) {
  hipLaunchKernelGGL((gemm_bf16_w4_kernel<EPI_BF16, 4, ABL, SCHED, DMA0>), dim3(n_tiles), dim3(256), 0, s, X, W, bias, out, K, ldx,
                     ldw, ldo, tiles_n, n_tiles);
  PG_HIP(hipGetLastError());
  return 0;
}

// M, N multiples of 256; K a multiple of 64.  abl > 0: micro-benchmark variants (bf16 epilogue only)
int launch_gemm_w4(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldx,
                   int ldw, int ldo, int epi, int abl) {
  const int tiles_m = M / 256, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  if (M % 256 || N % 256 || K % 64 || K < 64 || n_tiles < 1) return fail(1, "gemm_w4: shape");
  switch (abl) {
    case 0: case 11: break;
    case 1: return launch_w4_abl<1>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 2: return launch_w4_abl<2>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 3: return launch_w4_abl<3>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 4: return launch_w4_abl<4>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 5: return launch_w4_abl<5>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 6: return launch_w4_abl<0, 1, 0>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);    // X-pair-major MFMA order
    case 7: return launch_w4_abl<0, 0, 16>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);   // DMA pieces behind the fragment reads
    case 8: return launch_w4_abl<0, 2, 0>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);    // wave-staggered DMA slots
    case 10: return launch_w4_abl<10>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);        // DMA never waited for (timing)
    // 32x32x16 MFMA kernel: 11 = real (all epilogues below), 12.. its ablations
    case 12: return launch_w4m32_abl<1>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);      // no DMA
    case 13: return launch_w4m32_abl<2>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);      // no MFMA
    case 14: return launch_w4m32_abl<4>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);      // no epilogue
    case 15: return launch_w4m32_abl<5>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);      // no barrier
    case 16: return launch_w4m32_abl<0, 0>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);   // DMA pieces beside the fragment reads
    case 17: return launch_w4m32_abl<10>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);     // DMA never waited for
    default: return fail(1, "gemm_w4: unknown ablation");
  }
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  const int gm = gm_env ? gm_env : (K >= 4096 ? 2 : 4);
  dim3 grid(n_tiles), block(256);
  static const int mf_env = [] { const char* e = getenv("PGIBBS_GEMM_MFMA"); return e ? atoi(e) : 16; }();
  if (abl == 11 || mf_env == 32) {
#define PG_W4_CASE32(E)                                                                                                    \
  case E:                                                                                                                  \
    if (gm == 2) hipLaunchKernelGGL((gemm_bf16_w4m32_kernel<E, 2, 0>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles); \
    else hipLaunchKernelGGL((gemm_bf16_w4m32_kernel<E, 4, 0>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);     \
    break;
    switch (epi) {
      PG_W4_CASE32(EPI_BF16)
      PG_W4_CASE32(EPI_BF16_GELU)
      PG_W4_CASE32(EPI_F32_RESID)
      PG_W4_CASE32(EPI_F32)
      PG_W4_CASE32(EPI_F32_GELU)
      default:
        return fail(1, "gemm_w4: bad epilogue");
    }
#undef PG_W4_CASE32
    PG_HIP(hipGetLastError());
    return 0;
  }
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
