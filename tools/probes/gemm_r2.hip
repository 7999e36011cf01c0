// PROBE, not part of libpgibbs.so (round 4, VERDICT r03 item 1a: built, measured, lost; numbers in
// profiles/r04_two_resident_gemm_and_stagger.txt, analysis in DESIGN.md section 4 "Round 4").  Bit-identical with the 8-wave
// 256 x 256 kernel at 7 shapes, and its read-modify-write epilogue does hide under the other resident's main loop (fc2: 12 us of
// it visible) -- but a 256 x 128 tile needs 1.5 x the L2 -> LDS bytes per FLOP of a 256 x 256 tile and its main loop runs at
// 870-1000 TFLOP/s instead of 1240: out-projection 279 vs 260 us, fc2 1004 vs 767 us, config-2 iteration 96.3 vs 87.9 ms.  Two
// resident 256 x 256 tiles do not fit the register file (2 x 256 KB of accumulators = all of it).
// To try it again: copy next to csrc/gemm_w16.hip, add it to the Makefile, declare launch_gemm_r2 in kernels.h, give
// gemm_tail_tile64 a STAGES template parameter (4 stages = 64 KB) and route EPI_F32_RESID to it in launch_gemm_big with
// big_geometry(M, N, K, tile_n = 128, slots = 2) (git show HEAD~1 has the wiring).
// Residual-update GEMM with TWO resident workgroups per CU:   x[M][N] += A[M][K] . W[N][K]^T + bias[N]   (fp32 accumulate)
//
// The attention out-projection and fc2 of every layer behind `self.model.model(batch)` (/root/reference/src/pgen/esm_sampler.py:223;
// ESM-MSA-1b: row / column out-projections and fc2, esm_msa_sampler.py:236) end in a read-modify-write of the fp32 residual stream:
// 338 MB read + 338 MB written per launch at config 2.  In the 256 x 256 kernels (one workgroup per CU, 128 KB of LDS) that
// epilogue runs strictly AFTER the tile's main loop on the same CU -- 21 us of memory latency behind 31 us of MFMAs in the
// out-projection (0.33 of the MFMA peak, 0.52 of HBM; neither roof).  Here a tile is 256 token rows x 128 output
// features, held by 8 waves with 64 accumulator registers each, in 72 KB of LDS: two workgroups fit a CU (4 waves per SIMD at
// <= 128 VGPRs), they are scheduled independently, and one's read-modify-write epilogue runs under the other's main loop.
//
//   * K is walked in half-steps of 32 (one v_mfma_f32_16x16x32_bf16 deep): a slot is 256 A rows + 128 W rows of 64 B = 24 KB =
//     24 LDS-DMA pieces of 1 KiB (16 rows x 64 B, the ping-pong kernel's piece); a ring of THREE slots, two half-steps in flight
//     behind a counted s_waitcnt vmcnt(3); one s_barrier per half-step (it publishes the landed pieces of step h and retires
//     everybody's reads of the slot that step h + 2 overwrites).
//   * 16-B chunks of a 64-B row XOR-swizzled with pi[(row >> 2) & 3], pi = {0, 3, 2, 1}, on the DMA source address and on the
//     ds_read_b128 address (gemm_bf16.hip): conflict-free fragment reads.
//   * wave tile 64 x 64 (4 x 4 MFMA blocks): 8 fragment reads per 16 MFMAs, as the 16-wave kernel.
//   * same MFMA instruction, operand roles and k order as every other tile kernel, the bias added before the residual: a row's
//     result is bit-identical whichever kernel a batch / shard size selects (tests/test_gpu_kernels.py, shard tests).
//   * epilogue in two passes of 128 token rows through 64 KB of the (then idle) ring, leaving as whole 512-B rows; the residual
//     rows of pass 1 are in flight while pass 0 is added and stored.
//   * rows beyond the last full round of tiles (2 per CU) run as 64 x 64 tail tiles in front of the same grid (gemm_epilogue.h).
#include <stdlib.h>

#include "gemm_epilogue.h"

namespace pg {

constexpr int R2_SLOT = 24 * 1024;

// ABL (micro-benchmark ablations): 0 real kernel; 4 no epilogue (accumulators kept alive); 13 one half-step only (prologue + epilogue)
template <int GM, int ABL>
__global__ __launch_bounds__(512, 4) void gemm_bf16_r2_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ out, int K,
                                                             int ldx, int ldw, int ldo, int tiles_n, int n_tiles, int n_tail,
                                                             int tail_m0, int stagger_cycles) {
  __shared__ __attribute__((aligned(16))) char smem[3 * R2_SLOT];

  if (ABL == 0 && (int)blockIdx.x < n_tail) {
    // workgroup b runs on XCD b % 8: all column tiles of a 64-row block go to one XCD whenever the row blocks divide by 8
    const int tn64 = tiles_n * 2, bt = blockIdx.x, n_rb = n_tail / tn64;
    int rb, tn;
    if ((n_rb & 7) == 0) { const int j = bt >> 3; rb = (j / tn64) * 8 + (bt & 7); tn = j % tn64; }
    else { rb = bt / tn64; tn = bt % tn64; }
    gemm_tail_tile64<8, EPI_F32_RESID, false, 4>(X, W, bias, out, K, ldx, ldw, ldo, tail_m0 + rb * 64, tn * 64, smem);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 3, wn = wave >> 2;          // wave tile: A rows wm*64 .., W rows wn*64 ..

  int bid = blockIdx.x - n_tail;
  // experiment: the workgroups that take the second slot of each CU in the first dispatch round start late, so that the two
  // residents of a CU are out of phase (one's epilogue under the other's main loop) from the first tile on
  if (stagger_cycles > 0 && ((blockIdx.x >> 8) & 1) && blockIdx.x < 512) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)stagger_cycles) __builtin_amdgcn_s_sleep(32);
  }
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
  const int m0 = tile_m * 256, n0 = tile_n * 128;

  // ---- LDS-DMA: pieces 0-15 of a slot are A rows 16p .. 16p+15, pieces 16-23 W rows; wave w stages A pieces 2w, 2w+1 and W
  // piece w.  lane -> row (lane >> 2), LDS chunk (lane & 3) <- source chunk (lane & 3) ^ pi[(row >> 2) & 3]
  const int schunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)(m0 + wave * 32) * ldx), 0, (31 * ldx + K) * 2, 0x00020000);
  const rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)(n0 + wave * 16) * ldw), 0, (15 * ldw + K) * 2, 0x00020000);
  const int voff_x = ((lane >> 2) * ldx + schunk * 8) * 2;
  const int voff_w = ((lane >> 2) * ldw + schunk * 8) * 2;
  const int piece2 = 16 * ldx * 2;
  const int nh = K / 32;                            // half-steps; >= 2 (launcher)

  auto dma = [&](int h, int slot) {
    char* dst = smem + slot * R2_SLOT;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, PG_LDS_PTR(dst + (2 * wave) * 1024), 16, voff_x, h * 64, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, PG_LDS_PTR(dst + (2 * wave + 1) * 1024), 16, voff_x, h * 64 + piece2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, PG_LDS_PTR(dst + 16384 + wave * 1024), 16, voff_w, h * 64, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);          // row*64 + swizzled chunk*16
  const int xoff = wm * 4 * 1024 + foff;
  const int woff = 16384 + wn * 4 * 1024 + foff;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  dma(0, 0);
  dma(1, 1);
  const int nh_run = ABL == 13 ? 1 : nh;
  for (int h0 = 0; h0 < nh_run; h0 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int h = h0 + u;
      if (h >= nh_run) break;                        // wave-uniform
      // my pieces of half-step h have landed; those of h + 1 stay in flight
      if (h + 1 < nh) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                  // everybody's have; everybody is done reading slot (u + 2) % 3 (step h - 1)
      __builtin_amdgcn_sched_barrier(0);
      if (h + 2 < nh) dma(h + 2, (u + 2) % 3);
      const char* sb = smem + u * R2_SLOT;
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(sb + woff + i * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(sb + xoff + j * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (ABL == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }

  // ---- epilogue.  acc[i][j][r] = D[n = wn*64 + i*16 + fq*4 + r][m = wm*64 + j*16 + fr].  Pass p stages the token rows
  // wm*64 + p*32 + (0..31) of every wave: staged row sr = wm*32 + jj*16 + fr (128 rows of 512 B, 16-B chunk c = n / 4 XOR-swizzled
  // with sr & 31); wave w then owns the staged rows 16w .. 16w+15 = token rows m0 + (w >> 1)*64 + p*32 + (w & 1)*16 + (0..15) and
  // moves them out two whole rows per instruction (lanes 0-31 / 32-63).
  __syncthreads();                                  // every wave is done with the operand ring
  auto stage = [&](int p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 b4 = *(const float4*)(bias + n0 + wn * 64 + i * 16 + fq * 4);
      const int c = wn * 16 + i * 4 + fq;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int sr = wm * 32 + jj * 16 + fr;
        const f32x4 a = acc[i][p * 2 + jj];
        *(float4*)(smem + sr * 512 + ((c ^ (sr & 31)) << 4)) = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
      }
    }
  };
  const int half = lane >> 5, c16 = lane & 31;
  const int rstep2 = 2 * ldo * 4, voff = half * ldo * 4 + c16 * 16;
  auto rsrc_of = [&](int p) {
    return __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(m0 + (wave >> 1) * 64 + p * 32 + (wave & 1) * 16) * ldo + n0, 0, 0x7fffffff, 0x00020000);
  };
  const rsrc_t rs0 = rsrc_of(0), rs1 = rsrc_of(1);
  f32x4 r0[8], r1[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) r0[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff + it * rstep2, 0, 2));
  __builtin_amdgcn_sched_barrier(0);
  stage(0);
  __builtin_amdgcn_sched_barrier(0);                // r1 must not be hoisted above the staging (register budget)
#pragma unroll
  for (int it = 0; it < 8; ++it) r1[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, voff + it * rstep2, 0, 2));
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int sr = wave * 16 + it * 2 + half;
    const f32x4 v = *(const f32x4*)(smem + sr * 512 + ((c16 ^ (sr & 31)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r0[it] + v), rs0, voff + it * rstep2, 0, 2);
  }
  __syncthreads();
  stage(1);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int sr = wave * 16 + it * 2 + half;
    const f32x4 v = *(const f32x4*)(smem + sr * 512 + ((c16 ^ (sr & 31)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r1[it] + v), rs1, voff + it * rstep2, 0, 2);
  }
}

// x[M + tail_rows][N] += X . W^T + bias: M rows (a multiple of 256, may be 0) as 256 x 128 tiles, then tail_rows rows (a multiple
// of 64) as 64 x 64 tail tiles in front of the same grid.  N a multiple of 128, K of 32, K >= 64.
int launch_gemm_r2(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, float* out, int M, int N, int K, int ldx,
                   int ldw, int ldo, int abl, int tail_rows) {
  const int tiles_m = M / 256, tiles_n = N / 128, n_tiles = tiles_m * tiles_n;
  const int n_tail = (tail_rows / 64) * (N / 64), tail_m0 = M;
  if (M % 256 || N % 128 || K % 32 || K < 64 || tail_rows % 64 || n_tiles + n_tail < 1) return fail(1, "gemm_r2: shape");
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  static const int stagger = [] { const char* e = getenv("PGIBBS_R2_STAGGER"); return e ? atoi(e) : 0; }();
  const int gm = gm_env ? gm_env : (K >= 4096 ? 2 : 4);
  dim3 grid(n_tiles + n_tail), block(512);
#define PG_R2_ARGS X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, n_tail, tail_m0, stagger
  if (abl == 4) hipLaunchKernelGGL((gemm_bf16_r2_kernel<4, 4>), dim3(n_tiles), block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, 0, 0, 0);
  else if (abl == 13) hipLaunchKernelGGL((gemm_bf16_r2_kernel<4, 13>), dim3(n_tiles), block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, 0, 0, 0);
  else if (gm == 1) hipLaunchKernelGGL((gemm_bf16_r2_kernel<1, 0>), grid, block, 0, s, PG_R2_ARGS);
  else if (gm == 2) hipLaunchKernelGGL((gemm_bf16_r2_kernel<2, 0>), grid, block, 0, s, PG_R2_ARGS);
  else if (gm == 8) hipLaunchKernelGGL((gemm_bf16_r2_kernel<8, 0>), grid, block, 0, s, PG_R2_ARGS);
  else hipLaunchKernelGGL((gemm_bf16_r2_kernel<4, 0>), grid, block, 0, s, PG_R2_ARGS);
#undef PG_R2_ARGS
  PG_HIP(hipGetLastError());
  return 0;
}

}  // namespace pg
