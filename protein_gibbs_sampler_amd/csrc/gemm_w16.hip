// 256x256 bf16 MFMA GEMM tile for gfx950 with SIXTEEN waves per workgroup (four per SIMD), each owning a 64 x 64 block of the
// tile:   out[M][N] (+)= X[M][K] . W[N][K]^T + bias[N]   (fp32 accumulate)
//
// Same op, operand layout and k order as gemm_bf16.hip (every dense layer behind `self.model.model(batch)`,
// /root/reference/src/pgen/esm_sampler.py:223).  Where the 8-wave ping-pong kernel hand-schedules two wave groups against
// each other and the 4-wave kernel (tools/probes/gemm_w4.hip) software-pipelines inside one wave per SIMD, this one leaves the overlap to
// the hardware: four waves per SIMD at <= 128 VGPRs, a plain loop per wave
//     wait for my DMA pieces of K-step t -> s_barrier -> issue the pieces of K-step t+1 -> 2 x (8 fragment reads, 16 MFMAs)
// and the SIMD's scheduler runs one wave's MFMAs under the others' LDS reads, DMA issue and barrier waits.  The price is LDS
// read traffic (8 fragment reads per 16 MFMAs: 256 KB per K-step and CU against 192 / 128 KB of the 8- / 4-wave tiles).
//
//   * K-steps of 64: a slot is 256 X rows + 256 W rows of 128 B = 64 KB = 64 DMA pieces of 1 KiB (8 full 128-B rows per
//     wave-instruction, so every line of X / W passes the L2 -> L1 path once); two slots; wave w stages pieces 4w .. 4w+3.
//   * 16-B chunks of a row XOR-swizzled with (row & 7) (source-side permutation; conflict-free ds_read_b128).
//   * one s_barrier per K-step: it publishes the landed pieces of step t and retires everybody's reads of step t-1's slot.
//   * same k order and MFMA instruction as every other tile kernel -> bit-identical results for any split of a batch.
//   * grouped, XCD-aware tile order and LDS-staged epilogues as in the other 256 x 256 kernels (gemm_epilogue.h).
// Measured and removed again (QKV shape, this kernel 0.551-0.560 ms, the 8-wave ping-pong kernel 0.571-0.579): no barrier at all
// (timing only) 0.537; DMA never waited for 0.583 (= real: latency is covered); DMA pieces behind the first fragment reads, s_setprio
// around the MFMA clusters: no change; the barrier moved between the reads and the MFMAs of the second half-step (so that a released
// wave still has 16 MFMAs queued): 0.560-0.567; a 16-wave port of the two-group ping-pong schedule (4 barriers per K-step): 0.582;
// waves 8-15 skewed half a K-step behind waves 0-7 with still one barrier per K-step (their barrier between the reads and MFMAs of the
// second half-step, so that one group issues MFMAs from registers while the other restarts with LDS reads): 0.546 vs 0.552.
#include <stdlib.h>

#include "gemm_epilogue.h"

PG_OPS_BEGIN

constexpr int W16_KSLOT = 64 * 1024;

// tools/probes/gemm_clock.hip: shader-clock and 100 MHz wall-clock stamps around the main loop (compiled out of the library)
#ifdef PG_W16_PROF
__device__ unsigned long long* pg_w16_prof;      // [workgroup][4]: shader clock / wall clock at loop entry and exit
#define PG_W16_T(i)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0) {                                                                           \
      pg_w16_prof[(size_t)blockIdx.x * 4 + (i)] = __builtin_readcyclecounter();                       \
      pg_w16_prof[(size_t)blockIdx.x * 4 + (i) + 1] = __builtin_amdgcn_s_memrealtime();               \
    }                                                                                                 \
  } while (0)
#else
#define PG_W16_T(i)
#endif

// ABL (micro-benchmark ablations): 0 real kernel; 1 no LDS-DMA in the loop (slot 0 reused); 2 no MFMA; 4 no epilogue;
// 5 no barrier (timing only); 10 DMA never waited for (timing only)
template <int EPI, int GM, int ABL>
__global__ __launch_bounds__(1024, 1) void gemm_bf16_w16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                               const float* __restrict__ bias, void* __restrict__ out, int K,
                                                               int ldx, int ldw, int ldo, int tiles_n, int n_tiles, int n_tail,
                                                               int tail_m0) {
  __shared__ __attribute__((aligned(16))) char smem[2 * W16_KSLOT];

  // the first n_tail workgroups: 64 x 64 tiles of the rows beyond the last full round of 256 x 256 tiles (gemm_epilogue.h)
  const int nt_abs = n_tail < 0 ? -n_tail : n_tail;       // n_tail < 0: the tail workgroups are the LAST of the grid
  if (ABL == 0 && nt_abs && (n_tail > 0 ? (int)blockIdx.x < nt_abs : (int)blockIdx.x >= n_tiles)) {
    const int tn64 = tiles_n * 4, bt = n_tail > 0 ? blockIdx.x : blockIdx.x - n_tiles;
    gemm_tail_tile64<16, EPI>(X, W, bias, out, K, ldx, ldw, ldo, tail_m0 + (bt / tn64) * 64, (bt % tn64) * 64, smem);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 3, wn = wave >> 2;           // wave tile: X rows wm*64 .., W rows wn*64 ..

  int bid = n_tail > 0 ? blockIdx.x - n_tail : blockIdx.x;
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

  // ---- LDS-DMA: a slot holds 64 pieces (0-31: X rows 8p .. 8p+7, 32-63: W rows); wave w stages pieces 4w .. 4w+3, i.e. 32
  // consecutive rows of X (w < 8) or W.  Buffer form: one lane offset, piece / k offsets in the scalar offset; num_records =
  // the wave's 32-row band, so the prefetch past the end of K reads zeros without touching memory.
  const bool stage_w = wave >= 8;
  const int ld_ = stage_w ? ldw : ldx;
  const bf16_t* src = (stage_w ? W + (size_t)n0 * ldw : X + (size_t)m0 * ldx) + (size_t)(wave & 7) * 32 * ld_;
  const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (31 * ld_ + K) * 2, 0x00020000);
  const int dma_voff = ((lane >> 3) * ld_ + ((lane & 7) ^ (lane >> 3)) * 8) * 2;
  const int piece_bytes = 8 * ld_ * 2;
  const int lds_piece0 = wave * 4 * 1024;
  const int nk = K / 64;

  auto dma_step = [&](int t) {                       // this wave's 4 pieces of K-step t
    char* dst = smem + (t & 1) * W16_KSLOT + lds_piece0;
    const int soff = t < nk ? t * 128 : 0x7f000000;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst + g * 1024), 16, dma_voff, soff + g * piece_bytes, 0, 0);
  };

  // fragment tile T (16 rows x 128 B = 2 KiB), half kk: lane reads row fr, chunk (kk*4 + fq) ^ (fr & 7)
  const int fr = lane & 15, fq = lane >> 4;
  const int foff0 = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const int foff1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
  const int xbase = wm * 4 * 2048;
  const int wbase = 32 * 1024 + wn * 4 * 2048;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  dma_step(0);
  PG_W16_T(0);
  for (int t = 0; t < nk; ++t) {
    if ((ABL != 1 && ABL != 10) || t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of K-step t have landed
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 5) __builtin_amdgcn_s_barrier();      // everybody's have; everybody is done reading slot (t+1)&1 (step t-1)
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 1) dma_step(t + 1);
    const char* sb = smem + ((ABL == 1 ? 0 : t) & 1) * W16_KSLOT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(sb + wbase + i * 2048 + fo);
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(sb + xbase + j * 2048 + fo);
      if (ABL != 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma_op16(wf[i], xf[j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wf[i]), "v"(xf[i]));
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the trailing zero-fill pieces, before the LDS is reused
  PG_W16_T(2);

  if (ABL == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  // acc[i][j] is D[n = wn*64 + i*16 + fq*4 + r][m = wm*64 + j*16 + fr]
  auto elem = [&](int e, int& m_loc, int& n_loc) -> f32x4 {
    m_loc = wm * 64 + (e & 3) * 16 + fr;
    n_loc = wn * 64 + (e >> 2) * 16 + fq * 4;
    return acc[e >> 2][e & 3];
  };
  tile256_epilogue<EPI, 16>(elem, smem, wave, lane, m0, n0, bias, out, ldo);
}

#ifndef PG_F16      /* the strict mode's split operands are bf16 pairs in either build */
// ---------------------------------------------------------------------------------------------------------------------
// Strict precision mode: the three split-bf16 products of a projection from ONE pass over the operands.
//   X3 [M][3K], W3 [N][3K]: per group of 32 columns X3 = [xl | xh | xh], W3 = [wh | wl | wh] (elementwise.hip store_row_bf16).
// A "K-step" here is one group: the first 64 values (128 B) of the group from each operand row -- xl, xh / wh, wl of 32
// columns -- land in the same 128-B LDS rows as a plain K-step of 64, so the DMA pieces, the swizzle and the fragment
// addressing are those of gemm_bf16_w16_kernel (half 0 of a row = xl / wh, half 1 = xh / wl); only the source stride per
// step (192 B instead of 128) and the MFMA list differ:   acc += wh.xl ; acc += wl.xh ; acc += wh.xh   (48 MFMAs per 64 KB
// staged instead of 32: the operand feed, not the matrix pipe, bounds these kernels).  Per accumulator that is exactly the
// sequence a plain bf16 GEMM over K' = 3K walks through on this layout -- the k order of every other tile kernel -- so the
// result is bit-identical with launch_gemm_bf16 on the same operands (tests/test_gpu_strict_kernels.py), and a batch split
// into shards that pick different kernels still computes the same logits.
template <int EPI, int GM>
__global__ __launch_bounds__(1024, 1) void gemm_split3_w16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                                 const float* __restrict__ bias, void* __restrict__ out, int K,
                                                                 int ldx, int ldw, int ldo, int tiles_n, int n_tiles, int n_tail,
                                                                 int tail_m0) {
  __shared__ __attribute__((aligned(16))) char smem[2 * W16_KSLOT];
  // the first n_tail workgroups: 64 x 64 tiles of the rows beyond the last full round of 256 x 256 tiles (gemm_epilogue.h)
  if ((int)blockIdx.x < n_tail) {
    const int out_cols = (EPI == EPI_SPLIT3_GELU || EPI == EPI_SPLIT2_GELU) ? ldo / 3 : ldo;
    const int tn64 = tiles_n * 4, bt = blockIdx.x;
    (void)out_cols;
    gemm_tail_tile64<16, EPI, true>(X, W, bias, out, K, ldx, ldw, ldo, tail_m0 + (bt / tn64) * 64, (bt % tn64) * 64, smem);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  int bid = blockIdx.x - n_tail;
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

  // K = logical depth (columns of the unsplit operand); ldx, ldw = 3K-wide rows
  const bool stage_w = wave >= 8;
  const int ld_ = stage_w ? ldw : ldx;
  const bf16_t* src = (stage_w ? W + (size_t)n0 * ldw : X + (size_t)m0 * ldx) + (size_t)(wave & 7) * 32 * ld_;
  const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (31 * ld_ + 3 * K) * 2, 0x00020000);
  const int dma_voff = ((lane >> 3) * ld_ + ((lane & 7) ^ (lane >> 3)) * 8) * 2;
  const int piece_bytes = 8 * ld_ * 2;
  const int lds_piece0 = wave * 4 * 1024;
  const int nk = K / 32;                             // groups of 32 columns

  auto dma_step = [&](int t) {
    char* dst = smem + (t & 1) * W16_KSLOT + lds_piece0;
    const int soff = t < nk ? t * 192 : 0x7f000000;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst + g * 1024), 16, dma_voff, soff + g * piece_bytes, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int foff0 = fr * 128 + ((fq ^ (fr & 7)) << 4);            // xl / wh: chunks 0-3 of the row
  const int foff1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);      // xh / wl: chunks 4-7
  const int xbase = wm * 4 * 2048;
  const int wbase = 32 * 1024 + wn * 4 * 2048;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  dma_step(0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    dma_step(t + 1);
    const char* sb = smem + (t & 1) * W16_KSLOT;
    bf16x8 xl[4], xh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xl[j] = *(const bf16x8*)(sb + xbase + j * 2048 + foff0);
      xh[j] = *(const bf16x8*)(sb + xbase + j * 2048 + foff1);
    }
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) {                 // two W row blocks at a time: 48 fragment registers live, not 64
      bf16x8 wh[2], wl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        wh[u] = *(const bf16x8*)(sb + wbase + (ip * 2 + u) * 2048 + foff0);
        wl[u] = *(const bf16x8*)(sb + wbase + (ip * 2 + u) * 2048 + foff1);
      }
      // per accumulator: wh.xl, then wl.xh, then wh.xh -- three sweeps over the 8 accumulators, so no MFMA follows its producer
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ip * 2 + u][j] = mfma_op16(wh[u], xl[j], acc[ip * 2 + u][j]);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ip * 2 + u][j] = mfma_op16(wl[u], xh[j], acc[ip * 2 + u][j]);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ip * 2 + u][j] = mfma_op16(wh[u], xh[j], acc[ip * 2 + u][j]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  auto elem = [&](int e, int& m_loc, int& n_loc) -> f32x4 {
    m_loc = wm * 64 + (e & 3) * 16 + fr;
    n_loc = wn * 64 + (e >> 2) * 16 + fq * 4;
    return acc[e >> 2][e & 3];
  };
  tile256_epilogue<EPI, 16>(elem, smem, wave, lane, m0, n0, bias, out, ldo);
}

// X3 [M][3K], W3 [N][3K] in the split operand layout; K = logical depth (a multiple of 32); M, N multiples of 256.
// epi: EPI_F32, EPI_F32_RESID, EPI_SPLIT3_GELU or EPI_SPLIT2_GELU.
int launch_gemm_split3_w16(hipStream_t s, const bf16_t* X3, const bf16_t* W3, const float* bias, void* out, int M, int N, int K,
                           int ldo, int epi) {
  if (M % 256 || N % 256 || K % 32 || K < 32 || M < 256) return fail(1, "gemm_split3_w16: shape");
  // whole rounds of 256 x 256 tiles + 64 x 64 tail tiles for the rows of a mostly empty last round, as launch_gemm_big
  int m_main = M / 256, tail_rows = 0;
  gemm_big_geometry(M, N, 3 * K, &m_main, &tail_rows);
  const int tiles_m = m_main, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  const int n_tail = (tail_rows / 64) * (N / 64), tail_m0 = m_main * 256;
  const int gm = K >= 4096 ? 2 : 4;
  dim3 grid(n_tiles + n_tail), block(1024);
#define PG_S3_CASE(E)                                                                                                          \
  case E:                                                                                                                      \
    if (gm == 2) hipLaunchKernelGGL((gemm_split3_w16_kernel<E, 2>), grid, block, 0, s, X3, W3, bias, out, K, 3 * K, 3 * K, ldo, tiles_n, n_tiles, n_tail, tail_m0); \
    else hipLaunchKernelGGL((gemm_split3_w16_kernel<E, 4>), grid, block, 0, s, X3, W3, bias, out, K, 3 * K, 3 * K, ldo, tiles_n, n_tiles, n_tail, tail_m0);     \
    break;
  switch (epi) {
    PG_S3_CASE(EPI_F32)
    PG_S3_CASE(EPI_F32_RESID)
    PG_S3_CASE(EPI_SPLIT3_GELU)
    PG_S3_CASE(EPI_SPLIT2_GELU)
    default:
      return fail(1, "gemm_split3_w16: bad epilogue");
  }
#undef PG_S3_CASE
  PG_HIP(hipGetLastError());
  return 0;
}

#endif  // !PG_F16

template <int ABL>
static int launch_w16_abl(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int K, int ldx, int ldw,
                          int ldo, int tiles_n, int n_tiles) {
  hipLaunchKernelGGL((gemm_bf16_w16_kernel<EPI_BF16, 4, ABL>), dim3(n_tiles), dim3(1024), 0, s, X, W, bias, out, K, ldx, ldw, ldo,
                     tiles_n, n_tiles, 0, 0);
  PG_HIP(hipGetLastError());
  return 0;
}

// M, N multiples of 256; K a multiple of 64.  abl > 0: micro-benchmark variants (bf16 epilogue only)
int launch_gemm_w16(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldx,
                    int ldw, int ldo, int epi, int abl, int tail_rows) {
  const int tiles_m = M / 256, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  static const int tail_last = [] { const char* e = getenv("PGIBBS_GEMM_TAIL_LAST"); return e ? atoi(e) : 0; }();
  const int n_tail_abs = (tail_rows / 64) * (N / 64), tail_m0 = M;
  const int n_tail = tail_last ? -n_tail_abs : n_tail_abs;
  if (M % 256 || N % 256 || K % 64 || K < 64 || tail_rows % 64 || n_tiles + n_tail_abs < 1) return fail(1, "gemm_w16: shape");
  switch (abl) {
    case 0: break;
    case 1: return launch_w16_abl<1>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 2: return launch_w16_abl<2>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 4: return launch_w16_abl<4>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 5: return launch_w16_abl<5>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    case 10: return launch_w16_abl<10>(s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles);
    default: return fail(1, "gemm_w16: unknown ablation");
  }
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  const int gm = gm_env ? gm_env : (K >= 4096 ? 2 : 4);
  dim3 grid(n_tiles + n_tail_abs), block(1024);
  note_kernel("w16-256x256", n_tiles);
  if (n_tail_abs) note_kernel("tail64", n_tail_abs);
#define PG_W16_CASE(E)                                                                                                     \
  case E:                                                                                                                  \
    if (gm == 2) hipLaunchKernelGGL((gemm_bf16_w16_kernel<E, 2, 0>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, n_tail, tail_m0); \
    else hipLaunchKernelGGL((gemm_bf16_w16_kernel<E, 4, 0>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, n_tail, tail_m0);     \
    break;
  switch (epi) {
    PG_W16_CASE(EPI_BF16)
    PG_W16_CASE(EPI_BF16_GELU)
    PG_W16_CASE(EPI_F32_RESID)
    PG_W16_CASE(EPI_F32)
    PG_W16_CASE(EPI_F32_GELU)
    default:
      return fail(1, "gemm_w16: bad epilogue");
  }
#undef PG_W16_CASE
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
