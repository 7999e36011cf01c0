// Tile-height ladder of the 8-wave ping-pong GEMM (round 6):  out[M][N] (+)= X[M][K] . W[N][K]^T + bias
//
// The dense layers of the forward the reference reaches through `self.model.model(batch)` (/root/reference/src/pgen/esm_sampler.py:223;
// chains are independent there, :223-234, which is what lets one GPU run a 1/N shard of the batch).
//
// Why: a shard of BASELINE config 3 gives its residual GEMMs (N = 1280: five 256-column tiles per row panel) far too few tiles to
// quantise well on 256 CUs -- 32 chains: 165 tiles of 256 rows (one round, 64 % of the chip) or 220 of 192; 128 chains: 645 tiles =
// 2.52 rounds -- and fc1 of a 32-chain shard is 660 tiles = 2.58 rounds.  gemm_bf16_pp_kernel knows two heights (256, 192).  This
// file is the same main loop with the token rows of a tile as (XJ0 + XJ1) x 16 for ANY split of the 16-row blocks over the two
// wave groups (XJ0 = blocks of waves 0-3, XJ1 = blocks of waves 4-7, XJ0 - XJ1 in {0, 1}): heights 160 ... 240 in steps of 16.
//   32 chains (8256 rows): out-proj / fc2 as 235 tiles of 176 rows (one round, 92 % of the chip, 0.69 of a 256-row round instead
//   of 0.75), fc1 as 740 tiles of 224 rows (three nearly full rounds = 2.63 round-equivalents instead of 3);
//   128 chains: fc2 as 740 tiles of 224 rows (2.63 instead of 3).
// A row's products are accumulated in the same k order by the same MFMA as in every other tile kernel, so a row's bits do not
// depend on the tile height: shards stay bit-identical with the whole batch (tests/test_gpu_kernels.py, test_gpu_fullsize.py).
//
// Differences from gemm_bf16_pp_kernel (gemm_bf16.hip), whose comments explain the ping-pong schedule, the LDS ring of four half-K
// buffers and the swizzle:
//   * X staging: the (XJ0 + XJ1) 16-row pieces of a half-step are dealt round-robin to waves 0-3 (piece p -> wave p & 3), so a
//     wave stages 3 or 4 (2 or 3 ...) pieces and waits with the matching vmcnt; waves 4-7 stage the four W pieces each, as before.
//   * the two wave groups run the main loop with their own accumulator count (a wave-uniform branch at the top when XJ0 != XJ1:
//     same barriers in both bodies).
//   * epilogue: phase g stages the rows of wave group g through LDS ((XJg x 16) rows x 1 KiB <= 128 KB) and all eight waves move
//     them out as whole rows (2 XJg rows per wave) -- residual read-modify-write with the row loads of both phases issued before the
//     first store, or fc1's bias + GELU + 16-bit pack.
#include <stdio.h>
#include <stdlib.h>

#include "gemm_epilogue.h"

PG_OPS_BEGIN

namespace ladder {

#define PGL_AUX 2                                                /* buffer ops: nt (as the ping-pong kernel's epilogues) */
__device__ __forceinline__ rsrc_t row_rsrc(void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ f32x4 buf_load_f32x4(rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, PGL_AUX));
}
// the row step stays in the VGPR offset (gemm_bf16.hip: with an SGPR soffset the last dword of a 128-bit store was corrupted)
__device__ __forceinline__ void buf_store_f32x4(f32x4 v, rsrc_t rs, int voff, int row_off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, voff + row_off, 0, PGL_AUX);
}

constexpr int HALF_BYTES = 512 * 64;              // one half-buffer: (256 + 256) rows x 64 B

// s_waitcnt vmcnt(halves * npc): at most `halves` of this wave's half-steps of DMA still in flight (npc pieces each, wave-uniform)
#define PGL_WAIT(halves, lgkm)                                                                   \
  do {                                                                                           \
    if (npc == 4) asm volatile("s_waitcnt vmcnt(%0)" lgkm ::"n"((halves) * 4) : "memory");       \
    else if (npc == 3) asm volatile("s_waitcnt vmcnt(%0)" lgkm ::"n"((halves) * 3) : "memory");  \
    else asm volatile("s_waitcnt vmcnt(%0)" lgkm ::"n"((halves) * 2) : "memory");                \
  } while (0)

// stage the pieces of half-step (t, kk) this wave owns
__device__ __forceinline__ void stage_half(char* smem, const bf16_t* gsrc, size_t piece_stride, int lds_piece0, int lds_stride, int npc,
                                           int t, int kk) {
  char* hb = smem + ((t & 1) * 2 + kk) * HALF_BYTES + lds_piece0;
  const bf16_t* g = gsrc + (size_t)t * 64 + kk * 32;
  __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g), PG_LDS_PTR(hb), 16, 0, 0);
  __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g + piece_stride), PG_LDS_PTR(hb + lds_stride), 16, 0, 0);
  if (npc >= 3) __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g + 2 * piece_stride), PG_LDS_PTR(hb + 2 * lds_stride), 16, 0, 0);
  if (npc >= 4) __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g + 3 * piece_stride), PG_LDS_PTR(hb + 3 * lds_stride), 16, 0, 0);
}

// ---- epilogues: phase g = the XJg x 16 rows of wave group g, staged as fp32 rows of 1 KiB (256 columns), chunks XOR-swizzled with
// the staged row so that the fragment-shaped writes and the row-shaped reads are conflict-free (as epilogue_256) ----
template <int XJ>
__device__ __forceinline__ void stage_rows(f32x4 (&acc)[4][XJ], char* smem, int wn, int fr, int fq, const float* __restrict__ bias, int n0) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 b4 = *(const float4*)(bias + n0 + wn * 64 + i * 16 + fq * 4);
    const int c = wn * 16 + i * 4 + fq;                         // 16-B chunk of the 1-KB row
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int sr = j * 16 + fr;
      const f32x4 a = acc[i][j];
      *(float4*)(smem + sr * 1024 + ((c ^ (sr & 63)) << 4)) = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
    }
  }
}

// out += tile (fp32 residual stream): x_old + (acc + bias) per element, as epilogue_256's residual branch computes it
template <int XJ, int XJ0, int XJ1>
__device__ __forceinline__ void epilogue_resid(f32x4 (&acc)[4][XJ], char* smem, int grp, int wn, int wave, int lane, int m0, int n0,
                                               const float* __restrict__ bias, void* __restrict__ out, int ldo, int row_lo) {
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int RP0 = 2 * XJ0, RP1 = 2 * XJ1;                   // rows a wave moves in phase 0 / 1
  __syncthreads();
  const rsrc_t rs0 = row_rsrc((float*)out + (size_t)(m0 + wave * RP0) * ldo + n0);
  const rsrc_t rs1 = row_rsrc((float*)out + (size_t)(m0 + XJ0 * 16 + wave * RP1) * ldo + n0);
  const int rstep = ldo * 4, voff = lane * 16;
  f32x4 r0[RP0], r1[RP1];
#pragma unroll
  for (int it = 0; it < RP0; ++it) r0[it] = buf_load_f32x4(rs0, voff, it * rstep);
  __builtin_amdgcn_sched_barrier(0);
  if (grp == 0) stage_rows<XJ>(acc, smem, wn, fr, fq, bias, n0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < RP1; ++it) r1[it] = buf_load_f32x4(rs1, voff, it * rstep);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RP0; ++it) {
    const int sr = wave * RP0 + it;
    const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
    if (m0 + sr >= row_lo) buf_store_f32x4(r0[it] + v, rs0, voff, it * rstep);      // wave-uniform: a shifted last tile skips the previous tile's rows
  }
  __syncthreads();
  if (grp == 1) stage_rows<XJ>(acc, smem, wn, fr, fq, bias, n0);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < RP1; ++it) {
    const int sr = wave * RP1 + it;
    const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
    if (m0 + XJ0 * 16 + sr >= row_lo) buf_store_f32x4(r1[it] + v, rs1, voff, it * rstep);
  }
}

// fc1: bias in the staging, erf-GELU (the packed routine every tile kernel uses) + 16-bit pack on the way out, two rows of 512 B
// per wave instruction
template <int XJ, int XJ0, int XJ1>
__device__ __forceinline__ void epilogue_gelu16(f32x4 (&acc)[4][XJ], char* smem, int grp, int wn, int wave, int lane, int m0, int n0,
                                                const float* __restrict__ bias, void* __restrict__ out, int ldo) {
  const int fr = lane & 15, fq = lane >> 4;
  const int c8 = lane & 31;                                     // 8 consecutive features = fp32 chunks 2*c8, 2*c8+1
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (p) __syncthreads();
    if (grp == p) stage_rows<XJ>(acc, smem, wn, fr, fq, bias, n0);
    __syncthreads();
    const int xjp = p ? XJ1 : XJ0;                              // rows per wave in this phase: 2 xjp, two per iteration
    bf16_t* ob = (bf16_t*)out + (size_t)(m0 + (p ? XJ0 * 16 : 0) + wave * 2 * xjp) * ldo + n0;
#pragma unroll
    for (int it = 0; it < (XJ0 > XJ1 ? XJ0 : XJ1); ++it) {
      if (it < xjp) {
        const int r2 = it * 2 + (lane >> 5), sr = wave * 2 * xjp + r2;
        const float4 a = *(const float4*)(smem + sr * 1024 + (((2 * c8) ^ (sr & 63)) << 4));
        const float4 b = *(const float4*)(smem + sr * 1024 + (((2 * c8 + 1) ^ (sr & 63)) << 4));
        uint4 v;
        v.x = gelu_bf16out_pack2(a.x, a.y);
        v.y = gelu_bf16out_pack2(a.z, a.w);
        v.z = gelu_bf16out_pack2(b.x, b.y);
        v.w = gelu_bf16out_pack2(b.z, b.w);
        PG_NT_STORE((uint4*)(ob + (size_t)r2 * ldo + c8 * 8), v);
      }
    }
  }
}

// main loop + epilogue of one wave with XJ 16-row blocks (gemm_bf16_pp_kernel's loop, see there)
template <int EPI, int XJ, int XJ0, int XJ1>
__device__ __forceinline__ void wave_body(char* smem, const bf16_t* gsrc, size_t piece_stride, int lds_piece0, int lds_stride, int npc,
                                          int nk, int grp, int wn, int wave, int lane, int m0, int n0,
                                          const float* __restrict__ bias, void* __restrict__ out, int ldo, int row_lo) {
  f32x4 acc[4][XJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < XJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);          // row*64 + swizzled chunk*16
  const int xoff = (grp ? XJ0 * 16 : 0) * 64 + foff;
  const int woff = 256 * 64 + (wn * 64) * 64 + foff;
  bf16x8 wf[4], xf[XJ];
  for (int t = 0; t < nk; ++t) {
    const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // ---------------- L segment: fragments of half-step s = (t, kk) + DMA of half-step s+3 ----------------
      const char* hb = smem + ((t & 1) * 2 + kk) * HALF_BYTES;
      const bool issue = kk == 0 ? has1 : has2;
      if (issue) {
        if (kk == 0) stage_half(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, t + 1, 1);
        else stage_half(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, t + 2, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(hb + woff + i * 1024);
#pragma unroll
      for (int j = 0; j < XJ; ++j) xf[j] = *(const bf16x8*)(hb + xoff + j * 1024);
      // half-step s+1 must have landed before the barrier; s+2 and s+3 may stay in flight
      if (issue) {
        PGL_WAIT(2, " lgkmcnt(0)");
      } else if (kk == 1 && has1) {
        PGL_WAIT(1, " lgkmcnt(0)");                                   // (t+1,0) needed, (t+1,1) in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- C segment: 4 XJ MFMAs ----------------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < XJ; ++j)
          acc[i][j] = mfma_op16(wf[i], xf[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count (group 1 ran one extra up front)
  if constexpr (EPI == EPI_F32_RESID) epilogue_resid<XJ, XJ0, XJ1>(acc, smem, grp, wn, wave, lane, m0, n0, bias, out, ldo, row_lo);
  else epilogue_gelu16<XJ, XJ0, XJ1>(acc, smem, grp, wn, wave, lane, m0, n0, bias, out, ldo);
}

template <int EPI, int GM, int XJ0, int XJ1>
__global__ __launch_bounds__(512) void gemm_bf16_ppx_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                           const float* __restrict__ bias, void* __restrict__ out, int K, int ldx,
                                                           int ldw, int ldo, int tiles_n, int n_tiles, int m_rows) {
  static_assert(XJ0 >= XJ1 && XJ0 - XJ1 <= 1 && XJ0 <= 8 && XJ1 >= 4, "tile heights 128 ... 256 in steps of 16 rows");
  static_assert(EPI == EPI_F32_RESID || EPI == EPI_BF16_GELU, "ladder kernel: residual and fc1 epilogues");
  __shared__ __attribute__((aligned(16))) char smem[4 * HALF_BYTES];
  constexpr int NP = XJ0 + XJ1;                    // 16-row X pieces per half-step
  constexpr int TM = NP * 16;                      // token rows per tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;                       // 0: leads, 1: lags by one barrier
  const int wn = wave & 3;                         // wave tile: rows of group grp, W rows wn*64 ..

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
  // The last row panel may reach past the m_rows rows that exist (m_rows is a multiple of 16, >= TM): it is shifted up to end at
  // m_rows exactly, recomputes some of the previous panel's rows (same bits) and -- residual epilogue -- does not store them again.
  const int row_lo = tile_m * TM;
  const int m0 = row_lo + TM > m_rows ? m_rows - TM : row_lo;
  const int n0 = tile_n * 256;

  // ---- LDS-DMA source addressing.  Waves 4-7: W pieces (wave & 3) * 4 .. + 3 (16 rows each), contiguous.  Waves 0-3: X pieces
  // w, w + 4, w + 8, ... < NP.  lane -> row (lane >> 2), LDS chunk (lane & 3), swizzled on the source side.
  const bool stage_w = wave >= 4;
  const bf16_t* src = stage_w ? W : X;
  const int lds_ = stage_w ? ldw : ldx;
  const int srow0 = (stage_w ? n0 + wn * 64 : m0 + wn * 16) + (lane >> 2);
  const int schunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const bf16_t* gsrc = src + (size_t)srow0 * lds_ + schunk * 8;
  const size_t piece_stride = (size_t)(stage_w ? 16 : 64) * lds_;          // X: the wave's next piece is 4 pieces = 64 rows on
  const int lds_piece0 = stage_w ? 256 * 64 + wn * 4 * 1024 : wn * 1024;   // byte offset inside a half-buffer
  const int lds_stride = stage_w ? 1024 : 4096;
  const int npc = stage_w ? 4 : (NP - wn + 3) / 4;                         // pieces of a half-step this wave stages (2 ... 4)

  const int nk = K / 64;                            // >= 2 (launcher)
  stage_half(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, 0, 0);
  stage_half(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, 0, 1);
  stage_half(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, 1, 0);
  PGL_WAIT(2, "");                                  // half-step 0 landed; 1 and 2 stay in flight
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();      // stagger the two groups by one barrier interval

  if constexpr (XJ0 == XJ1) {
    wave_body<EPI, XJ0, XJ0, XJ1>(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, nk, grp, wn, wave, lane, m0, n0, bias, out, ldo, row_lo);
  } else {
    if (grp == 0) wave_body<EPI, XJ0, XJ0, XJ1>(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, nk, grp, wn, wave, lane, m0, n0, bias, out, ldo, row_lo);
    else wave_body<EPI, XJ1, XJ0, XJ1>(smem, gsrc, piece_stride, lds_piece0, lds_stride, npc, nk, grp, wn, wave, lane, m0, n0, bias, out, ldo, row_lo);
  }
}
#undef PGL_WAIT

template <int EPI, int XJ0, int XJ1>
static int launch_h(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int K, int ldx, int ldw, int ldo,
                    int tiles_n, int n_tiles, int gm, int m_rows) {
  dim3 grid(n_tiles), block(512);
  if (gm == 2) hipLaunchKernelGGL((gemm_bf16_ppx_kernel<EPI, 2, XJ0, XJ1>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, m_rows);
  else hipLaunchKernelGGL((gemm_bf16_ppx_kernel<EPI, 4, XJ0, XJ1>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, m_rows);
  PG_HIP(hipGetLastError());
  return 0;
}

}  // namespace ladder

// Heights this build instantiates per epilogue (each is one more kernel per operand flavour and tile grouping)
bool gemm_ladder_has(int epi, int h) {
  if (epi == EPI_F32_RESID) return h == 160 || h == 176 || h == 208 || h == 224 || h == 240;
  if (epi == EPI_BF16_GELU) return h == 208 || h == 224 || h == 240;
  return false;
}

// ceil(m_live / h) row panels of h token rows cover rows [0, m_live); m_rows (a multiple of 16, >= h, >= m_live) rows exist in X
// and out -- the last panel is shifted up to end at m_rows when it would reach past them.  N a multiple of 256, K of 64, K >= 128
int launch_gemm_ladder(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int m_live, int m_rows, int h,
                       int N, int K, int ldx, int ldw, int ldo, int epi) {
  if (!gemm_ladder_has(epi, h) || m_live < 1 || m_rows < h || m_rows < m_live || m_rows % 16 || N % 256 || K % 64 || K < 128)
    return fail(1, "gemm_ladder: shape / height / epilogue");
  const int tiles_m = (m_live + h - 1) / h;
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  const int gm = gm_env == 2 || gm_env == 4 ? gm_env : (K >= 4096 ? 2 : 4);
  const int tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  char label[32];
  snprintf(label, sizeof label, "ppx%dx256", h);
  note_kernel(label, n_tiles);
#define PGL_ARGS s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, gm, m_rows
  if (epi == EPI_F32_RESID) {
    switch (h) {
      case 160: return ladder::launch_h<EPI_F32_RESID, 5, 5>(PGL_ARGS);
      case 176: return ladder::launch_h<EPI_F32_RESID, 6, 5>(PGL_ARGS);
      case 208: return ladder::launch_h<EPI_F32_RESID, 7, 6>(PGL_ARGS);
      case 224: return ladder::launch_h<EPI_F32_RESID, 7, 7>(PGL_ARGS);
      default: return ladder::launch_h<EPI_F32_RESID, 8, 7>(PGL_ARGS);
    }
  }
  switch (h) {
    case 208: return ladder::launch_h<EPI_BF16_GELU, 7, 6>(PGL_ARGS);
    case 224: return ladder::launch_h<EPI_BF16_GELU, 7, 7>(PGL_ARGS);
    default: return ladder::launch_h<EPI_BF16_GELU, 8, 7>(PGL_ARGS);
  }
#undef PGL_ARGS
}

PG_OPS_END
