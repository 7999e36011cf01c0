// Full-row residual GEMM with the following LayerNorm in its epilogue, for d_model = 768 (ESM-MSA-1b; round 6):
//   x[M][768] += A[M][K] . W[768][K]^T + bias ;   h[M][768] = LayerNorm(x; gamma, beta)  as 16-bit operand rows
// These are the two attention out-projections of every AxialTransformerLayer -- row_self_attention / column_self_attention out_proj
// followed by the next block's pre-LayerNorm -- inside the forward the reference reaches through
// `self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_msa_sampler.py:136,236).
//
// Why a tile that spans all 768 output columns: with 256-column tiles a token row's LayerNorm statistics are spread over three
// workgroups, so the row goes back to HBM (fp32 read-modify-write by the GEMM) and is read again by a LayerNorm kernel -- 4.0 + 2.4 GB
// per out-projection at config 4, both kernels HBM-bound (LayerNorm is 9.4 % of the iteration).  A tile of 112 token rows x 768
// columns OWNS its rows: the epilogue adds the residual, stores x, and normalises each row while it is still on the chip -- 4.85 GB,
// no cross-workgroup counter, no "last arriver" (what killed PGIBBS_LN_FUSE in round 3).
//
// Bit-identical with the unfused path by construction: a row's products are accumulated by the same MFMA in the same k order as
// in every other tile kernel, the residual add is x_old + (acc + bias) as in the ping-pong kernel's epilogue, and the LayerNorm is
// ln_row.h's ln_inplace / store_row_bf16 on one wave per row -- the stand-alone kernel's arithmetic on the same fp32 values.  So the
// dispatch may pick it by shape alone; shards and batch sizes may mix the two forms (tests/test_gpu_msa.py).
//
// CDNA4 mapping.  8 waves; wave w owns ALL 112 rows x output columns [96 w, 96 w + 96): 7 x 6 accumulator fragments = 168 registers.
//   * W is wave-PRIVATE in LDS: each wave stages the 96 weight rows it multiplies (6 pieces of 16 rows x 64 B per half-step of k = 32)
//     into its own two slots -- no barrier is needed to recycle them, only the wave's own reads.  W (1.2 MB at K = 768) stays in L2.
//   * the activation rows are shared: 7 pieces per half-step, one per wave 0..6, in a ring of four slots (they come from HBM: three
//     half-steps of lead).  LDS-DMA (`global_load_lds`, 16 B per lane), XOR swizzle on the source side, as the ping-pong kernel.
//   * the two wave groups (0-3 / 4-7; waves w and w + 4 share a SIMD) run one barrier apart: while one group's 42 MFMAs own the matrix
//     pipe the other issues DMA and reads fragments.  Per half-step and wave: read 7 + 6 fragments, issue W(s+2) into the slot just read
//     and X(s+3), wait until W(s+1) / X(s+1) have landed (counted vmcnt: X(s+2), W(s+2), X(s+3) stay in flight), barrier, MFMAs, barrier.
//   * epilogue in three phases (row blocks {0,1}, {2,3}, {4,5,6}: 32 / 32 / 48 rows x 3 KB of LDS): every wave stages acc + bias for
//     its 96 columns, then wave w takes rows w, w + 8, ...: x_old (loaded before the staging) + staged row -> x (fp32, streamed) ->
//     ln_inplace -> 16-bit row of h, optionally in column-major token order (the fused column QKV + attention kernel's operand).
#include <stdlib.h>

#include <type_traits>

#include "gemm_epilogue.h"
#include "ln_row.h"

PG_OPS_BEGIN

namespace rowln {

constexpr int XJ = 7;                      // 16-row blocks of token rows per tile
constexpr int TM = XJ * 16;                // 112 token rows
constexpr int NB = 6;                      // 16-column blocks per wave
constexpr int N = 768;                     // output columns = d_model
constexpr int XSLOT = TM * 64;             // one half-step of activation rows: 112 x 64 B
constexpr int WSLOT = 96 * 64;             // one half-step of a wave's 96 weight rows
constexpr int W_BASE = 4 * XSLOT;          // X ring: 4 slots; then per wave 2 W slots
constexpr int MAIN_BYTES = W_BASE + 8 * 2 * WSLOT;      // 126 976
constexpr int EPI_BYTES = 48 * N * 4;                   // 147 456: 48 staged rows of 3 KB
static_assert(MAIN_BYTES <= EPI_BYTES && EPI_BYTES <= 160 * 1024, "LDS budget");

// at most `n` of this wave's DMA instructions still in flight (n is a compile-time immediate)
#define PGR_WAIT(n, lgkm) asm volatile("s_waitcnt vmcnt(%0)" lgkm ::"n"(n) : "memory")

// ABL (tools/rowln_bench.py only): 0 = the kernel; 1 = main loop only (nothing stored); 2 = four half-steps of main loop + the epilogue
template <int ABL>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ h, int K, int lda, int ldw, int m_live, int m_rows,
                                                        float eps, int cm_R, int cm_C, int n_first, int n_pop, int stagger_ticks) {
  __shared__ __attribute__((aligned(16))) char smem[EPI_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;                       // 0: leads, 1: lags by one barrier
  // De-phasing.  Every tile is a compute phase (main loop: L2 -> LDS feed, no HBM to speak of) followed by an HBM phase (the epilogue
  // moves 0.86 MB per tile), and all tiles take the same time: left alone, the 256 CUs run their main loops together with HBM idle
  // and then their epilogues together at 6.5 TB/s -- 0.61 + 0.75 ms per launch at config 4, one after the other
  // (profiles/r06_rowln_bench.txt).  The workgroups of the FIRST round (one per CU) therefore start in n_pop populations,
  // population p waiting p x stagger before its first tile; later workgroups inherit the phase of the CU they land on, so at any
  // time 1 / n_pop of the chip is in its HBM phase and the rest computes.  A scheduling hint only: results cannot depend on it.
  if (n_pop > 1 && (int)blockIdx.x < n_first) {
    const int pop = ((int)blockIdx.x >> 3) % n_pop;          // workgroup b runs on XCD b & 7: populations alternate inside every XCD
    if (pop) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      const unsigned long long wait = (unsigned long long)pop * (unsigned)stagger_ticks;
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
  }
  // the last row panel may reach past the m_rows rows that exist: shifted up to end there; rows below row_lo belong to its neighbour
  const int row_lo = blockIdx.x * TM;
  const int m0 = row_lo + TM > m_rows ? m_rows - TM : row_lo;

  // ---- LDS-DMA addressing: lane -> row (lane >> 2) of a 16-row piece, 16-B chunk (lane & 3), swizzled on the source side
  const int schunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const bf16_t* wsrc = W + (size_t)(wave * 96 + (lane >> 2)) * ldw + schunk * 8;      // + p * 16 rows, + k
  const size_t wpiece = (size_t)16 * ldw;
  const bool has_x = wave < XJ;                                                         // waves 0..6 stage one activation piece each
  const bf16_t* xsrc = A + (size_t)(m0 + (has_x ? wave : 0) * 16 + (lane >> 2)) * lda + schunk * 8;
  char* const wring = smem + W_BASE + wave * 2 * WSLOT;
  auto issue_w = [&](int s) {
    char* dst = wring + (s & 1) * WSLOT;
    const bf16_t* g = wsrc + (size_t)s * 32;
#pragma unroll
    for (int p = 0; p < NB; ++p) __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g + p * wpiece), PG_LDS_PTR(dst + p * 1024), 16, 0, 0);
  };
  auto issue_x = [&](int s) {
    if (has_x) __builtin_amdgcn_global_load_lds(PG_GLB_PTR(xsrc + (size_t)s * 32), PG_LDS_PTR(smem + (s & 3) * XSLOT + wave * 1024), 16, 0, 0);
  };

  f32x4 acc[NB][XJ];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < XJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nhs = ABL >= 2 ? 4 : K / 32;           // half-steps; >= 4 (launcher).  ABL 12 / 13 / 14 (timing only): 2 + no LayerNorm arithmetic / no residual row loads / no global stores
  // prologue: W(0) X(0) W(1) X(1) X(2) in this order, so that the counted waits below always leave X(s+2), W(s+2), X(s+3) in flight
  issue_w(0);
  issue_x(0);
  issue_w(1);
  issue_x(1);
  issue_x(2);
  if (has_x) PGR_WAIT(8, ""); else PGR_WAIT(6, "");       // W(0), X(0) landed
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();     // stagger the two groups by one barrier interval

  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);          // row*64 + swizzled chunk*16
  bf16x8 wf[NB], xf[XJ];
  for (int s = 0; s < nhs; ++s) {
    // ---------------- L segment ----------------
    const char* xs = smem + (s & 3) * XSLOT + foff;
    const char* ws = wring + (s & 1) * WSLOT + foff;
#pragma unroll
    for (int i = 0; i < NB; ++i) wf[i] = *(const bf16x8*)(ws + i * 1024);
#pragma unroll
    for (int j = 0; j < XJ; ++j) xf[j] = *(const bf16x8*)(xs + j * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // my W slot is free again
    __builtin_amdgcn_sched_barrier(0);
    const bool iw = s + 2 < nhs, ix = s + 3 < nhs;
    if (iw) issue_w(s + 2);
    if (ix) issue_x(s + 3);
    // W(s+1), X(s+1) must have landed before the barrier; X(s+2), W(s+2), X(s+3) may stay in flight
    if (ix) { if (has_x) PGR_WAIT(8, ""); else PGR_WAIT(6, ""); }
    else if (iw) { if (has_x) PGR_WAIT(7, ""); else PGR_WAIT(6, ""); }
    else PGR_WAIT(0, "");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- C segment: 42 MFMAs ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        acc[i][j] = mfma_op16(wf[i], xf[j], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count
  __syncthreads();
  if (ABL == 1) {
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < XJ; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }

  // ---------------- epilogue: three phases of row blocks {0,1} {2,3} {4,5,6} ----------------
  // Per phase: (1) every wave stages acc + bias of the phase's row blocks for its 96 columns; (2) the NEXT phase's residual rows are
  // requested (the accumulators just staged are dead: their registers hold the loads), so only the first phase waits for HBM;
  // (3) wave w finishes rows w, w + 8, ...: x_old + staged row -> x (fp32, streamed), ln_inplace, 16-bit row of h.
  auto load_x = [&](auto J0c, auto RPWc, float4 (&xo)[decltype(RPWc)::value][3]) {
    constexpr int J0 = decltype(J0c)::value, RPW = decltype(RPWc)::value;
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
      const float4* x4 = (const float4*)(x + (size_t)(m0 + J0 * 16 + wave + it * 8) * N);
#pragma unroll
      for (int c = 0; c < 3; ++c) xo[it][c] = ABL == 13 ? make_float4(1.f, 2.f, 3.f, (float)lane) : x4[lane + 64 * c];
    }
  };
  // stage acc + bias: lane holds D[n = 96 w + 16 i + 4 fq + r][row = 16 j + fr]; chunk (float4) index of the 3-KB row = 24 w + 4 i + fq,
  // XOR-swizzled with the staged row inside its group of 64 chunks (fragment-shaped writes and row-shaped reads both conflict-free)
  auto stage = [&](auto J0c, auto J1c) {
    constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = wave * 24 + i * 4 + fq;
      const float4 b4 = *(const float4*)(bias + wave * 96 + i * 16 + fq * 4);
#pragma unroll
      for (int j = J0; j < J1; ++j) {
        const int sr = (j - J0) * 16 + fr;
        const f32x4 a = acc[i][j];
        *(float4*)(smem + sr * (N * 4) + (((c & ~63) | ((c ^ sr) & 63)) << 4)) = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
      }
    }
  };
  auto finish = [&](auto J0c, auto RPWc, float4 (&v)[decltype(RPWc)::value][3]) {
    constexpr int J0 = decltype(J0c)::value, RPW = decltype(RPWc)::value;
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
      const int sr = wave + it * 8;
      const int g = m0 + J0 * 16 + sr;                                     // token row (wave-uniform)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4 t = *(const float4*)(smem + sr * (N * 4) + ((c * 64 + ((lane ^ sr) & 63)) << 4));
        v[it][c] = make_float4(v[it][c].x + t.x, v[it][c].y + t.y, v[it][c].z + t.z, v[it][c].w + t.w);
      }
      if (g >= row_lo) {                                                   // else: the neighbouring panel's row (shifted last panel)
        float4* xr = (float4*)(x + (size_t)g * N);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (ABL == 14) asm volatile("" ::"v"(v[it][c].x), "v"(v[it][c].y), "v"(v[it][c].z), "v"(v[it][c].w));
          else PG_NT_STORE(xr + lane + 64 * c, v[it][c]);
        }
      }
    }
    // the stand-alone LayerNorm kernel's own routine, row by row (ln_row.h: the ONLY definition of the arithmetic -- an interleaved
    // restatement of it, `ln_inplace_rows`, was measured to differ in the last bit of some h values and to buy nothing: the
    // LayerNorm arithmetic is free here, profiles/r06_rowln_epilogue_ablation.txt)
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
      const int g = m0 + J0 * 16 + wave + it * 8;
      if (g >= row_lo && g < m_live) {
        float4 w8[kMaxCh];
#pragma unroll
        for (int c = 0; c < 3; ++c) w8[c] = v[it][c];
        if (ABL != 12) ln_inplace(w8, N / 4, lane, N, eps, gamma, beta);
        size_t hrow = (size_t)g;
        if (cm_R > 0) {                                                    // token row (b R + r) C + c -> operand row (b C + c) R + r
          const int br = g / cm_C, cc = g - br * cm_C, bb = br / cm_R, rr = br - bb * cm_R;
          hrow = ((size_t)bb * cm_C + cc) * cm_R + rr;
        }
        if (ABL == 14) { asm volatile("" ::"v"(w8[0].x), "v"(w8[1].y), "v"(w8[2].z)); }
        else store_row_bf16(h + hrow * N, w8, N / 4, lane);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  using I6 = std::integral_constant<int, 6>;
  using I7 = std::integral_constant<int, 7>;
  // (No prefetch of the next phase's residual rows: with 120 accumulator registers still live in the first phase the extra 48 spill,
  // and scratch traffic would break the counted vmcnt waits of the main loop -- the kernel must stay spill-free.)
  {
    float4 xa[4][3];
    load_x(I0{}, I4{}, xa);
    __builtin_amdgcn_sched_barrier(0);
    stage(I0{}, I2{});
    __syncthreads();
    finish(I0{}, I4{}, xa);
  }
  {
    float4 xb[4][3];
    load_x(I2{}, I4{}, xb);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    stage(I2{}, I4{});
    __syncthreads();
    finish(I2{}, I4{}, xb);
  }
  {
    float4 xc[6][3];
    load_x(I4{}, I6{}, xc);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    stage(I4{}, I7{});
    __syncthreads();
    finish(I4{}, I6{}, xc);
  }
}
#undef PGR_WAIT

}  // namespace rowln

// May the full-row kernel take this residual GEMM + LayerNorm?  d_model = 768, K a multiple of 32 with at least four half-steps; the
// operand rows [0, m_rows) must exist (m_rows a multiple of 16, >= 112).  K is capped: at K = 3072 (fc2) the 112-row tile's 2.2x
// L2 -> LDS bytes per FLOP make the main loop slower than the 256 x 256 tiles by more than the LayerNorm pass costs.
//
// MEASURED (round 6; profiles/r06_rowln_bench.txt, r06_rowln_epilogue_ablation.txt, r06_rowln_stagger.txt, r06_msa_cfg4_rowln_ab.txt) and
// OFF by default (PGIBBS_ROWLN=1 switches it on: tests/test_gpu_msa.py, tools/rowln_bench.py).  Bit-identical x and h, and at
// config 4 (526 336 rows, K = 768) 1.34-1.37 ms per launch against 0.86 + 0.40 = 1.26-1.28 ms for the two launches it replaces
// (iteration 146.3 against 143.3 ms; LayerNorm 13.6 -> 4.5 ms, the GEMM family +12.4 ms).  Why: the main loop alone is FAST (0.61 ms,
// 1010 TFLOP/s: wave-private W slots, no barrier on the W side) and the epilogue moves its 4.85 GB at 7-8 TB/s when it runs alone
// (0.68 ms; without the LayerNorm arithmetic the same, without the residual loads -0.24 ms, without the stores -0.30 ms) -- but the
// two run one after the other on every CU: with one workgroup per CU (168 accumulator registers per lane, 144 KB of LDS) nothing
// computes while a tile's rows stream, and a CU streams at ~25 GB/s whether or not HBM is busy (the same per-CU rate the 256-column
// kernel's read-modify-write epilogue and the persistent single-chain trunk's row loads run at), so de-phasing the CUs (2-4
// populations, 10-45 us apart) changes nothing (1.32-1.43 ms).  The unfused pair moves 32 % more bytes but its LayerNorm kernel keeps
// 8 workgroups per CU in flight.  A second resident workgroup would need half the accumulators (a 56-row tile: 2x the W traffic per
// FLOP) -- the trade round 4's 256 x 128 two-resident GEMM lost.  Closed.
bool gemm_rowln_ok(int m_rows, int N, int K) {
  static const int on = [] { const char* e = getenv("PGIBBS_ROWLN"); return e ? atoi(e) : 0; }();
  static const int kmax = [] { const char* e = getenv("PGIBBS_ROWLN_KMAX"); return e ? atoi(e) : 1024; }();
  return on && N == rowln::N && K % 32 == 0 && K >= 128 && K <= kmax && m_rows >= rowln::TM && m_rows % 16 == 0;
}

int launch_gemm_rowln(hipStream_t s, const bf16_t* A, const bf16_t* W, const float* bias, float* x, const float* gamma,
                      const float* beta, bf16_t* h, int m_live, int m_rows, int K, int lda, int ldw, float eps, int colmajor_R,
                      int colmajor_C, int abl) {
  if (K % 32 || K < 128 || m_rows < rowln::TM || m_rows % 16 || m_live < 1 || m_live > m_rows) return fail(1, "gemm_rowln: shape");
  const int tiles = (m_live + rowln::TM - 1) / rowln::TM;
  note_kernel("rowln112x768", tiles);
  // de-phasing of the first round (see the kernel): populations and the delay between them, in ticks of s_memrealtime (100 MHz)
  static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  static const int pop_env = [] { const char* e = getenv("PGIBBS_ROWLN_POP"); return e ? atoi(e) : 1; }();        // measured: no effect (1 = off)
  static const int us_env = [] { const char* e = getenv("PGIBBS_ROWLN_STAGGER_US"); return e ? atoi(e) : 0; }();
  const int n_pop = (tiles >= 2 * n_cu && pop_env >= 1 && pop_env <= 8) ? pop_env : 1;      // fewer than two rounds: nothing to overlap
  const int period_us = us_env > 0 ? us_env * n_pop : (int)(0.043 * K + 41.0);                 // ~ one tile: main loop + epilogue
  const int stagger_ticks = n_pop > 1 ? period_us * 100 / n_pop : 0;
#define PGR_ARGS dim3(tiles), dim3(512), 0, s, A, W, bias, x, gamma, beta, h, K, lda, ldw, m_live, m_rows, eps, colmajor_R, colmajor_C, n_cu, n_pop, stagger_ticks
  if (abl == 1) hipLaunchKernelGGL(rowln::gemm_rowln_kernel<1>, PGR_ARGS);
  else if (abl == 2) hipLaunchKernelGGL(rowln::gemm_rowln_kernel<2>, PGR_ARGS);
  else if (abl == 12) hipLaunchKernelGGL(rowln::gemm_rowln_kernel<12>, PGR_ARGS);
  else if (abl == 13) hipLaunchKernelGGL(rowln::gemm_rowln_kernel<13>, PGR_ARGS);
  else if (abl == 14) hipLaunchKernelGGL(rowln::gemm_rowln_kernel<14>, PGR_ARGS);
  else hipLaunchKernelGGL(rowln::gemm_rowln_kernel<0>, PGR_ARGS);
#undef PGR_ARGS
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
