// bf16 MFMA GEMM for gfx950:  out[M][N] (+)= X[M][K] . W[N][K]^T + bias[N]   (fp32 accumulate)
//
// This is the op behind every dense layer of the forward pass the reference runs through
// fair-esm (`self.model.model(batch)`, /root/reference/src/pgen/esm_sampler.py:223): QKV
// projection, attention out-projection, fc1 (+erf-GELU), fc2, LM-head dense.  SURVEY.md A.5:
// these GEMMs are 96.5 % of the FLOPs of a Gibbs iteration.
//
// Design (CDNA4, 64-wide waves):
//   * both operands are K-contiguous ([rows][K], PyTorch Linear layout), so W rows feed the MFMA
//     A operand and activation rows the B operand: D[n][m].  A lane then holds 4 consecutive n
//     for one token m.
//   * four kernels, picked by how many tiles they put on the 256 CUs (launch_gemm_bf16_variant):
//       - 256x256x64 ping-pong kernel (gemm_bf16_pp_kernel): 8 waves as two staggered groups, LDS-DMA ring of four
//         half-K buffers, LDS-staged coalesced epilogues with non-temporal stores; m-panels of a ragged last round of
//         tiles are peeled into a second, small launch;
//       - 128x128 and 64x64 lockstep double-buffered tiles (gemm_bf16_kernel) for 64 .. ~4000 token rows;
//       - weight-streaming kernel (gemm_bf16_skinny_kernel) up to 48 rows;
//       - deep-K residual GEMMs with few tiles run as parallel K-splits + a fixed-order reduction.
//     The tile kernels accumulate a row's products in the same k order, so large batches are bit-identical however the
//     rows are split over launches or GPUs (the weight-streaming and split-K paths of small batches differ, by rounding).
//   * HBM/L2 -> LDS by direct `global_load_lds` (16 B/lane, no VGPR round trip); 16-B LDS chunks are XOR-swizzled by
//     permuting the per-lane SOURCE address (the LDS image of a glds is lane-linear), so every ds_read_b128
//     lane group touches 16 distinct 16-B slots (conflict-free).
//   * 1-D grid remapped so each XCD (private L2) owns a contiguous range of tiles.
//   * epilogues fused: +bias, GELU, bf16 pack, fp32 residual read-modify-write.
#include <stdlib.h>

#include "gemm_epilogue.h"

PG_OPS_BEGIN

// Stage ROWS x 64 bf16 (128 B per row) into LDS.  One wave-instruction = 8 rows = 1 KiB, written
// lane-linearly; lane l covers row (l>>3), LDS chunk (l&7), which receives global chunk (l&7)^(row&7).
template <int ROWS, int NW>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ src, int ld, int row0, int k0, char* lds_tile,
                                           int wave, int lane) {
  const int rl = lane >> 3;
  const int c = (lane & 7) ^ rl;
#pragma unroll
  for (int i = 0; i < ROWS / 8 / NW; ++i) {
    const int rb = wave + i * NW;
    const bf16_t* g = src + (size_t)(row0 + rb * 8 + rl) * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g), PG_LDS_PTR(lds_tile + rb * 1024), 16, 0, 0);
  }
}

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_bf16_kernel(
    const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, const float* __restrict__ bias, void* __restrict__ out,
    int K, int ldx, int ldw, int ldo, int tiles_n, int n_tiles, long split_stride) {
  constexpr int NWM = BM / WM, NWN = BN / WN, NW = NWM * NWN;
  if (EPI == EPI_F32_PARTIAL) {                    // K-split blockIdx.y: its K range of both operands, its partial map
    X += (size_t)blockIdx.y * K;
    W += (size_t)blockIdx.y * K;
    out = (float*)out + (size_t)blockIdx.y * split_stride;
  }
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int TILE_BYTES = (BM + BN) * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % NWM, wn = wave / NWM;

  // XCD-aware bijective remap: block b runs on XCD b % 8; give each XCD a contiguous tile range.
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = n_tiles >> 3, r = n_tiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = (bid / tiles_n) * BM;
  const int n0 = (bid % tiles_n) * BN;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;
  stage_tile<BM, NW>(X, ldx, m0, 0, smem, wave, lane);
  stage_tile<BN, NW>(W, ldw, n0, 0, smem + BM * 128, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fr = lane & 15;        // fragment row within a 16-row MFMA tile
  const int fq = lane >> 4;        // k-chunk (8 bf16) within a 32-wide MFMA k-step
  const int sw = fr & 7;           // row & 7 (tile bases are multiples of 16)
  int cur = 0;
  for (int t = 0; t < nk; ++t) {
    char* xt = smem + cur * TILE_BYTES;
    char* wt = xt + BM * 128;
    if (t + 1 < nk) {
      char* nx = smem + (cur ^ 1) * TILE_BYTES;
      stage_tile<BM, NW>(X, ldx, m0, (t + 1) * 64, nx, wave, lane);
      stage_tile<BN, NW>(W, ldw, n0, (t + 1) * 64, nx + BM * 128, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = (((fq + 4 * kk) ^ sw) << 4);
      bf16x8 wf[TN], xf[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const bf16x8*)(wt + (wn * WN + i * 16 + fr) * 128 + coff);
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *(const bf16x8*)(xt + (wm * WM + j * 16 + fr) * 128 + coff);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = mfma_op16(wf[i], xf[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane holds D[n = nb + fq*4 + r][m = mb + fr], r = 0..3
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WN + i * 16 + fq * 4;
    const float4 b4 = EPI == EPI_F32_PARTIAL ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(bias + n);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * WM + j * 16 + fr;
      float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
      if (EPI == EPI_F32_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
      if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
        uint2 p;
        // every tile kernel sends the fc1 GELU through the same packed routine: rows stay bit-identical for any batch split
        p.x = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v0, v1) : pack_op2(v0, v1);
        p.y = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v2, v3) : pack_op2(v2, v3);
        *(uint2*)((bf16_t*)out + (size_t)m * ldo + n) = p;
      } else if (EPI == EPI_F32_RESID) {
        float4* o = (float4*)((float*)out + (size_t)m * ldo + n);
        float4 r = *o;
        r.x += v0; r.y += v1; r.z += v2; r.w += v3;
        *o = r;
      } else {
        *(float4*)((float*)out + (size_t)m * ldo + n) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}



// ------------------------------------------------------------------------------------------------
// Coalesced epilogue of the 256x256 kernels.  A lane's MFMA result is 4 consecutive output features of ONE token
// row, so direct stores are 32 scattered 8/16-byte pieces per lane: measured store-issue-bound (0.19 ms of a 0.79 ms
// QKV GEMM).  Instead the tile goes through the (now idle) 128 KB of LDS -- XOR-swizzled so both the fragment-shaped
// writes and the row-shaped reads are conflict-free -- and leaves as full rows: every wave-instruction stores
// 64 lanes x 16 B = 1 KiB of contiguous output (bf16: two 512-B rows; fp32: one 1-KB row; the fp32 residual
// read-modify-write uses the same row-shaped accesses, 16 loads in flight per lane).
// ------------------------------------------------------------------------------------------------
// The epilogue's row-shaped stores (and the residual rows it reads) are streamed with the non-temporal policy: a round of
// tiles writes 4 MB per XCD -- its whole L2 -- which otherwise evicts the X / W k-slices the main loops share through it.
// Measured at the four ESM-1b shapes: QKV 0.589 -> 0.580 ms, fc1 0.859 -> 0.818, out-proj 0.308 -> 0.285, fc2 0.856 -> 0.818;
// whole iteration 96.1 -> 94.1 ms.  (Non-temporal loads/stores in LayerNorm: no effect.)
#define PG_EPI_AUX 2                                             /* buffer ops: nt */
#define PG_EPI_STORE(p, v) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)(p))
__device__ __forceinline__ rsrc_t row_rsrc(void* base) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ f32x4 buf_load_f32x4(rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, PG_EPI_AUX));
}
// Stores keep the row step in the VGPR offset: with an SGPR soffset the compiler's hazard recogniser assumes a 128-bit
// store's data registers may be overwritten by the very next VALU instruction, and on gfx950 that corrupted the last
// dword of the stored row (seen as wrong .w components in lanes 12-15 of each 16) -- with soffset = 0 it pads the hazard.
__device__ __forceinline__ void buf_store_f32x4(f32x4 v, rsrc_t rs, int voff, int row_off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, voff + row_off, 0, PG_EPI_AUX);
}

template <int EPI, bool NO_STORE = false>
__device__ __forceinline__ void epilogue_256(f32x4 (&acc)[4][8], char* smem, int wm, int wn, int wave, int lane, int m0,
                                             int n0, const float* __restrict__ bias, void* __restrict__ out, int ldo) {
  const int fr = lane & 15, fq = lane >> 4;
  __syncthreads();
  // fp32-staged epilogues (fp32 outputs, and fc1's bf16+GELU).  Two phases; in phase p EVERY wave stages its accumulator
  // columns j = 4p..4p+3 (64 token rows per wave group -> 128 staged rows x 1 KiB = all of LDS), then wave w owns the 16
  // staged rows w*16.. = token rows m0 + (w>>2)*128 + (4p + (w&3))*16 + it and moves them out as whole 1-KiB rows.
  // Residual variant: the 16 row loads of a phase (64 VGPRs) are issued BEFORE that phase's LDS staging, and phase 1's
  // loads before phase 0's stores, so a tile exposes about one memory latency instead of four.
  if (EPI == EPI_BF16_GELU || EPI == EPI_F32 || EPI == EPI_F32_GELU || EPI == EPI_F32_RESID) {
    auto stage = [&](int p) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 b4 = *(const float4*)(bias + n0 + wn * 64 + i * 16 + fq * 4);
        const int c = wn * 16 + i * 4 + fq;                       // 16-B chunk of the 1-KB row
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int sr = wm * 64 + jj * 16 + fr;
          const f32x4 a = acc[i][p * 4 + jj];
          float4 v = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
          if (EPI == EPI_F32_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
          *(float4*)(smem + sr * 1024 + ((c ^ (sr & 63)) << 4)) = v;
        }
      }
    };
    // first token row of this wave's 16 staged rows in phase p (wave-uniform)
    auto grow = [&](int p) { return m0 + (wave >> 2) * 128 + (p * 4 + (wave & 3)) * 16; };
    if (EPI == EPI_BF16_GELU) {
      // fc1: the erf-GELU (about 20 VALU ops per element) runs in the LDS -> global phase, where it overlaps with the
      // store-issue stalls of the other wave on the SIMD.  The value rounded to bf16 is the same fp32 the direct
      // epilogue would produce.
      const int c8 = lane & 31;                                   // 8 consecutive features = fp32 chunks 2*c8, 2*c8+1
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (p) __syncthreads();
        stage(p);
        __syncthreads();
        bf16_t* ob = (bf16_t*)out + (size_t)grow(p) * ldo + n0;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int r2 = it * 2 + (lane >> 5), sr = wave * 16 + r2;
          const float4 a = *(const float4*)(smem + sr * 1024 + (((2 * c8) ^ (sr & 63)) << 4));
          const float4 b = *(const float4*)(smem + sr * 1024 + (((2 * c8 + 1) ^ (sr & 63)) << 4));
          uint4 v;
          v.x = gelu_bf16out_pack2(a.x, a.y);
          v.y = gelu_bf16out_pack2(a.z, a.w);
          v.z = gelu_bf16out_pack2(b.x, b.y);
          v.w = gelu_bf16out_pack2(b.z, b.w);
          PG_EPI_STORE((uint4*)(ob + (size_t)r2 * ldo + c8 * 8), v);
        }
      }
      return;
    }
    if (EPI == EPI_F32_RESID) {
      // Row-shaped accesses as buffer ops: wave-uniform row base in the resource, the row step in an SGPR offset, one
      // VGPR (lane*16) for all 64 accesses -- 64-bit per-row VGPR addresses would not leave room for 32 rows in flight.
      const rsrc_t rs0 = row_rsrc((float*)out + (size_t)grow(0) * ldo + n0);
      const rsrc_t rs1 = row_rsrc((float*)out + (size_t)grow(1) * ldo + n0);
      const int rstep = ldo * 4, voff = lane * 16;
      f32x4 r0[16], r1[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) r0[it] = buf_load_f32x4(rs0, voff, it * rstep);
      __builtin_amdgcn_sched_barrier(0);
      stage(0);
      __builtin_amdgcn_sched_barrier(0);          // r1 must not be hoisted above the staging (register budget)
#pragma unroll
      for (int it = 0; it < 16; ++it) r1[it] = buf_load_f32x4(rs1, voff, it * rstep);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int sr = wave * 16 + it;
        const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
        buf_store_f32x4(r0[it] + v, rs0, voff, it * rstep);
      }
      __syncthreads();
      stage(1);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int sr = wave * 16 + it;
        const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
        buf_store_f32x4(r1[it] + v, rs1, voff, it * rstep);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p) __syncthreads();
      stage(p);
      __syncthreads();
      float* ob = (float*)out + (size_t)grow(p) * ldo + n0;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int sr = wave * 16 + it;
        *(float4*)(ob + (size_t)it * ldo + lane * 4) = *(const float4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
      }
    }
    return;
  }
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 b4 = *(const float4*)(bias + n0 + wn * 64 + i * 16 + fq * 4);
      const int c = wn * 8 + i * 2 + (fq >> 1);                   // 16-B chunk of the 512-B row
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = wm * 128 + j * 16 + fr;
        const float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
        uint2 p;
        p.x = pack_op2(v0, v1);
        p.y = pack_op2(v2, v3);
        *(uint2*)(smem + row * 512 + ((c ^ (row & 31)) << 4) + (fq & 1) * 8) = p;
      }
    }
    __syncthreads();
    const int c = lane & 31;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = (wave * 16 + it) * 2 + (lane >> 5);
      const uint4 v = *(const uint4*)(smem + row * 512 + ((c ^ (row & 31)) << 4));
      if (NO_STORE) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); continue; }    // ablation: all but the global stores
      PG_EPI_STORE((uint4*)((bf16_t*)out + (size_t)(m0 + row) * ldo + n0 + c * 8), v);
    }
  }
}

// out += tile for the 192 x 256 tile (XJ = 6): a wave holds token rows wm*96 + j*16 + fr, j < 6.  Two phases of 96 staged rows x
// 1 KiB; in phase p every wave stages its columns j = 3p .. 3p+2 (48 token rows per wave group), then wave w owns the 12 staged
// rows w*12 .. = token rows m0 + (w>>2)*96 + p*48 + (w&3)*12 + it.  x_old + (acc + bias) per element, as epilogue_256's residual
// branch computes it; the same early row loads (phase 1's before phase 0's stores).
__device__ __forceinline__ void epilogue_192_resid(f32x4 (&acc)[4][6], char* smem, int wm, int wn, int wave, int lane, int m0,
                                                   int n0, const float* __restrict__ bias, void* __restrict__ out, int ldo) {
  const int fr = lane & 15, fq = lane >> 4;
  __syncthreads();
  auto stage = [&](int p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 b4 = *(const float4*)(bias + n0 + wn * 64 + i * 16 + fq * 4);
      const int c = wn * 16 + i * 4 + fq;                         // 16-B chunk of the 1-KB row
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int sr = wm * 48 + jj * 16 + fr;
        const f32x4 a = acc[i][p * 3 + jj];
        *(float4*)(smem + sr * 1024 + ((c ^ (sr & 63)) << 4)) = make_float4(a[0] + b4.x, a[1] + b4.y, a[2] + b4.z, a[3] + b4.w);
      }
    }
  };
  auto grow = [&](int p) { return m0 + (wave >> 2) * 96 + p * 48 + (wave & 3) * 12; };
  const rsrc_t rs0 = row_rsrc((float*)out + (size_t)grow(0) * ldo + n0);
  const rsrc_t rs1 = row_rsrc((float*)out + (size_t)grow(1) * ldo + n0);
  const int rstep = ldo * 4, voff = lane * 16;
  f32x4 r0[12], r1[12];
#pragma unroll
  for (int it = 0; it < 12; ++it) r0[it] = buf_load_f32x4(rs0, voff, it * rstep);
  __builtin_amdgcn_sched_barrier(0);
  stage(0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < 12; ++it) r1[it] = buf_load_f32x4(rs1, voff, it * rstep);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int sr = wave * 12 + it;
    const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
    buf_store_f32x4(r0[it] + v, rs0, voff, it * rstep);
  }
  __syncthreads();
  stage(1);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int sr = wave * 12 + it;
    const f32x4 v = *(const f32x4*)(smem + sr * 1024 + ((lane ^ (sr & 63)) << 4));
    buf_store_f32x4(r1[it] + v, rs1, voff, it * rstep);
  }
}

// ------------------------------------------------------------------------------------------------
// "Ping-pong" 256x256x64 kernel (the hot one).
//
// The 8 waves form two groups of four (waves w and w+4 share a SIMD).  Work on a K-tile is cut into
// two half-steps (k 0..31 / 32..63); a wave alternates between an L segment (12 ds_read_b128 of the
// next half-step's fragments + 4 LDS-DMA pieces of the NEXT K-tile) and a C segment (32 MFMAs), with a
// raw s_barrier after every segment.  Group 1 executes one extra barrier up front, so its segments
// are shifted by one: while one group's MFMAs own the matrix pipe, the other group is reading LDS and
// issuing DMA.  DMA completion is a counted `s_waitcnt vmcnt(4)` (never 0 in steady state): a piece
// issued in one L segment is only waited for at the end of the wave's next L segment, i.e. two barrier
// intervals later.
//
// LDS: 4 half-buffers [tile parity][k half], each 256 X rows + 256 W rows of 64 B (32 bf16).
// A half-buffer is rewritten (for tile t+1) only after both groups finished reading tile t-1 from it:
//   piece (t+1, kk) is issued in L(t, kk);  last read of (t-1, kk) is group 1's L(t-1, kk), which ends
//   at least one barrier earlier for every wave.
// 16-B chunks are XOR-swizzled with pi[(row>>2)&3], pi = {0,3,2,1} (applied to the DMA source address
// and to the ds_read address) so that every ds_read_b128 lane group hits 16 distinct 16-B slots.
// ------------------------------------------------------------------------------------------------
// ABL (micro-benchmark ablations only): 0 = real kernel, 1 = no LDS-DMA inside the K loop (tile 0 reused),
// 2 = no MFMA (fragments kept alive), 3 = no ds_read (fragments loaded once)
// XJ = 16-row blocks of token rows per wave: 8 = the 256 x 256 tile; 6 = a 192 x 256 tile (residual epilogue only), chosen by
// launch_gemm_big when 256-row tiles would leave a third of the CUs idle (165 tiles of a 32-chain shard's out-projection / fc2 on
// 256 CUs: 220 tiles of three quarters the work instead).  Same k order per accumulator: a row's bits do not depend on the tile.
constexpr int RS_ROUNDS = 1024;                   // rounds per XCD the pacing words cover (beyond: no pacing)
constexpr unsigned long long RS_LIMIT = 4000;     // give up after 40 us of waiting (s_memrealtime ticks of 10 ns)
template <int EPI, int ABL = 0, int GM = 4, int XJ = 8>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                          const float* __restrict__ bias, void* __restrict__ out, int K,
                                                          int ldx, int ldw, int ldo, int tiles_n, int n_tiles, int n_tail,
                                                          int tail_m0, unsigned* rsync) {
  constexpr int HALF_BYTES = 512 * 64;            // one half-buffer: (256 + 256) rows x 64 B
  __shared__ __attribute__((aligned(16))) char smem[4 * HALF_BYTES];
  // the first n_tail workgroups: 64 x 64 tiles of the rows beyond the last full round of 256 x 256 tiles (gemm_epilogue.h)
  // n_tail > 0: they are the FIRST workgroups of the grid; n_tail < 0: the LAST |n_tail| (PGIBBS_GEMM_TAIL_LAST)
  const int nt_abs = n_tail < 0 ? -n_tail : n_tail;
  if (ABL == 0 && nt_abs && (n_tail > 0 ? (int)blockIdx.x < nt_abs : (int)blockIdx.x >= n_tiles)) {
    // workgroup b runs on XCD b % 8: all column tiles of a 64-row block go to one XCD (they share the block's X rows through its
    // L2) whenever the row blocks divide by 8 (tail rows are multiples of 256: at least by 4)
    const int tn64 = tiles_n * 4, bt = n_tail > 0 ? blockIdx.x : blockIdx.x - n_tiles, n_rb = nt_abs / tn64;
    int rb, tn;
    if ((n_rb & 7) == 0) { const int j = bt >> 3; rb = (j / tn64) * 8 + (bt & 7); tn = j % tn64; }
    else { rb = bt / tn64; tn = bt % tn64; }
    gemm_tail_tile64<8, EPI>(X, W, bias, out, K, ldx, ldw, ldo, tail_m0 + rb * 64, tn * 64, smem);
    return;
  }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2;                      // 0: leads, 1: lags by one barrier
  const int wm = grp, wn = wave & 3;              // wave tile: rows wm*(16 XJ).. of X, rows wn*64.. of W
  static_assert(XJ == 8 || (XJ == 6 && EPI == EPI_F32_RESID && ABL == 0), "192-row tiles: residual epilogue only");
  constexpr int TM = XJ * 32;                     // token rows per tile
  constexpr int NPX = XJ / 2;                     // 16-row DMA pieces per X-staging wave and half-step (W-staging waves: 4)

  int bid = n_tail > 0 ? blockIdx.x - n_tail : blockIdx.x;
  // Round pacing (round 5; rsync != nullptr: deep-K residual GEMMs, fc2).  Workgroup b runs on XCD b % 8 and an XCD's 32 CUs take
  // its workgroups in order, so workgroups 32r .. 32r+31 of an XCD are its "round" r: the GM x n rectangle of tiles that is meant to
  // walk K together and share X / W k-slices through the XCD's 4 MB L2.  Only the first round does by itself -- afterwards every
  // CU starts its next tile whenever it finishes, the rectangle's tiles drift apart by more K-steps than the L2 holds (352 KB per
  // step, 80 steps at K = 5120) and each re-fetches its slices through the fabric: fc2 read 3.3 GB per launch against 1.2 GB for
  // walking in step, and a lone round of 255 tiles -- launch, fill and drain included -- took 150 us where a steady-state round
  // took 157 (profiles/r05_lockstep_rounds.txt).  With PGIBBS_GEMM_RSYNC=1 the workgroups of a round wait for each other before their
  // first DMA (an ablation: it removes 40 % of the reads and does not pay, see round_sync_words()).  This
  // is a scheduling hint, not a dependency: after RS_LIMIT the wait gives up (CUs taken by another process, a CU mask) and switches
  // the pacing off for good; results cannot depend on it.
  if (rsync && ABL == 0) {
    if (threadIdx.x == 0 && __hip_atomic_load(rsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      const int xcd = bid & 7, q = bid >> 3, round = q >> 5;
      const int per_xcd = (n_tiles >> 3) + (xcd < (n_tiles & 7) ? 1 : 0);
      const unsigned size = (unsigned)((per_xcd - round * 32) < 32 ? (per_xcd - round * 32) : 32);
      if (round < RS_ROUNDS && round > 0) {         // round 0 starts together anyway
        unsigned* a = rsync + 16 + ((xcd * RS_ROUNDS + round) << 1);
        __hip_atomic_fetch_add(a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool gave_up = false;
        while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < size) {
          __builtin_amdgcn_s_sleep(8);
          if (__builtin_amdgcn_s_memrealtime() - t0 > RS_LIMIT) { gave_up = true; break; }
        }
        if (gave_up) __hip_atomic_store(rsync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // re-arm for the next launch: whoever leaves last clears both words (nobody is waiting on them any more)
        if (__hip_atomic_fetch_add(a + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == size - 1u) {
          __hip_atomic_store(a, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(a + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  if (ABL == 18 && (bid & 7) != 0) return;        // timing experiment: only the workgroups of XCD 0 run (1/8 of the tiles)
  if (ABL == 19 && (bid & 7) > 1) return;         // ... XCDs 0 and 1
  {
    const int xcd = bid & 7, q = n_tiles >> 3, r = n_tiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // grouped rasterisation inside the XCD's range: GM m-panels x all n-tiles per group, m fastest, so the ~32
  // tiles an XCD runs concurrently form a GM x (32/GM) rectangle that shares X and W k-slices through its L2.
  int tile_m, tile_n;
  {
    const int tiles_m = n_tiles / tiles_n;
    const int gsz = GM * tiles_n, g = bid / gsz, within = bid - g * gsz;
    const int rows = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tile_m = g * GM + within % rows;
    tile_n = within / rows;
  }
  const int m0 = tile_m * TM;
  const int n0 = tile_n * 256;

  // ---- LDS-DMA source addressing: wave stages pieces wave*4 .. wave*4+3 of the 32 pieces (16 rows each) of a
  // half-tile; pieces 0-15 are X rows, 16-31 are W rows.  lane -> row (lane>>2), LDS chunk (lane&3).
  // (192-row tiles: the four X-staging waves take 3 pieces each, X rows 0..191 of the same LDS layout.)
  const bool stage_w = wave >= 4;
  const bf16_t* src = stage_w ? W : X;
  const int lds_ = stage_w ? ldw : ldx;
  const int srow0 = (stage_w ? n0 + (wave & 3) * 64 : m0 + (wave & 3) * (NPX * 16)) + (lane >> 2);
  const int schunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);                 // pi[(row>>2)&3] = (-g)&3
  const bf16_t* gsrc = src + (size_t)srow0 * lds_ + schunk * 8;            // + i*16 rows, + k
  const size_t piece_stride = (size_t)16 * lds_;
  const int lds_piece0 = stage_w ? 256 * 64 + (wave & 3) * 4 * 1024 : (wave & 3) * NPX * 1024;  // byte offset inside a half-buffer

  auto stage_pieces = [&](int t, int kk, int i0, int i1) {
    char* hb = smem + ((t & 1) * 2 + kk) * HALF_BYTES + lds_piece0;
    const bf16_t* g = gsrc + (size_t)(ABL == 17 ? 0 : t) * 64 + kk * 32;   // ABL 17: always K-tile 0 -> every DMA hits in L2
#pragma unroll
    for (int i = i0; i < i1; ++i)
      __builtin_amdgcn_global_load_lds(PG_GLB_PTR(g + i * piece_stride), PG_LDS_PTR(hb + i * 1024), 16, 0, 0);
  };
  auto stage_half = [&](int t, int kk) {
    if (XJ == 8 || stage_w) stage_pieces(t, kk, 0, 4); else stage_pieces(t, kk, 0, NPX);
  };
  // wait until at most `halves` of this wave's half-steps of DMA are still in flight (4 or NPX pieces each; wave-uniform branch)
#define PG_PP_WAIT(halves, lgkm)                                                                              \
  do {                                                                                                        \
    if (XJ == 8 || stage_w) asm volatile("s_waitcnt vmcnt(%0)" lgkm ::"n"((halves) * 4) : "memory");          \
    else asm volatile("s_waitcnt vmcnt(%0)" lgkm ::"n"((halves) * NPX) : "memory");                           \
  } while (0)

  f32x4 acc[4][XJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < XJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / 64;                            // >= 2 (launcher)
  stage_half(0, 0);
  stage_half(0, 1);
  stage_half(1, 0);
  PG_PP_WAIT(2, "");                                // half-step 0 landed; 1 and 2 stay in flight
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();      // stagger the two groups by one barrier interval

  const int fr = lane & 15, fq = lane >> 4;
  const int foff = fr * 64 + ((fq ^ ((0 - (fr >> 2)) & 3)) << 4);          // row*64 + swizzled chunk*16
  const int xoff = (wm * XJ * 16) * 64 + foff;
  const int woff = 256 * 64 + (wn * 64) * 64 + foff;

  bf16x8 wf[4], xf[XJ];
  for (int t = 0; t < (ABL == 13 ? 1 : nk); ++t) {          // ABL 13: one K-tile only -> times prologue + epilogue
    const bool has1 = (t + 1 < nk) && ABL != 13, has2 = (t + 2 < nk) && ABL != 13;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      // ---------------- L segment: fragments of half-step s = (t, kk) + DMA of half-step s+3 ----------------
      constexpr bool NO_DMA = (ABL == 1 || ABL == 8 || ABL == 9 || ABL == 11), NO_DS = (ABL == 3 || ABL == 8 || ABL == 9 || ABL == 11 || ABL == 12),
                     NO_BAR = (ABL == 9 || ABL == 11 || ABL == 12);
      const char* hb = smem + (((NO_DMA ? 0 : (t & 1)) * 2) + kk) * HALF_BYTES;
      const bool issue = kk == 0 ? has1 : has2;
      if (issue && !NO_DMA) {
        if (kk == 0) stage_half(t + 1, 1); else stage_half(t + 2, 0);
      }
      if (!NO_DS || t == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(hb + woff + i * 1024);
#pragma unroll
        for (int j = 0; j < XJ; ++j) xf[j] = *(const bf16x8*)(hb + xoff + j * 1024);
      }
      // half-step s+1 must have landed before the barrier; s+2 and s+3 (4 pieces each) may stay in flight
      if (NO_DMA) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (ABL == 15) {                                         // timing only: never wait for the DMA
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if (issue) {
        PG_PP_WAIT(2, " lgkmcnt(0)");
      } else if (kk == 1 && has1) {
        PG_PP_WAIT(1, " lgkmcnt(0)");                                 // (t+1,0) needed, (t+1,1) in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!NO_BAR) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- C segment: 32 MFMAs ----------------
      __builtin_amdgcn_s_setprio(1);
      if (ABL != 2 && ABL != 12) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < XJ; ++j)
            acc[i][j] = mfma_op16(wf[i], xf[j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wf[i]));
#pragma unroll
        for (int j = 0; j < XJ; ++j) asm volatile("" ::"v"(xf[j]));
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (!NO_BAR) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count
#undef PG_PP_WAIT

  if (ABL == 10 || ABL == 11 || ABL == 12) {     // ablation: keep the accumulators alive, store nothing
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < XJ; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if constexpr (XJ == 8) epilogue_256<EPI, ABL == 16>(acc, smem, wm, wn, wave, lane, m0, n0, bias, out, ldo);
  else epilogue_192_resid(acc, smem, wm, wn, wave, lane, m0, n0, bias, out, ldo);
}

// Measured alternatives that were NOT faster on MI355X and were removed again (QKV GEMM, M=66048 N=3840 K=1280, steady-state
// clocks -- time >= 300 launches: the first ~10 ms after idle run at ramping clocks and mislead):
//   * "v3" -- all waves in lockstep, DMA pieces and fragment prefetch interleaved after every 4 MFMAs, one barrier per
//     half-step; ping-pong on v_mfma_f32_32x32x16_bf16 (same MFMA busy cycles, more issue stalls);
//   * 1 or 2 of the 4 DMA pieces issued inside the MFMA segment: -1 % / -2 %;
//   * half of the first round's workgroups started half a tile late (to de-phase the chip-wide store bursts): 0 ... -2 %;
//   * a "W-stationary" tile order (4-5 n-tiles at a time over a band of 32 m-panels, so their W panels stay in the XCD's
//     L2 while X streams): 0 ... -2 % on all four shapes, although making EVERY DMA hit in L2 (ablation 17) is worth 14 %;
//   * direct 16-B stores from the accumulators (16 rows x 64-B segments per instruction, possible with a permuted W-row
//     to fragment assignment) instead of the LDS-staged 512-B rows: timing-only ablation 0.582 vs 0.573 ms;
//   * touching the residual tile's cache lines at the start of the main loop (so that the read half of the epilogue's
//     read-modify-write is spread out): out-proj 0.31 -> 0.35 ms;
//   * a persistent kernel (one workgroup per CU; after a tile, waves 0-3 store it from 64 KB of LDS staging while waves
//     4-7 issue the next tile's first three half-steps of DMA; the split is by wave because vmcnt counts stores too):
//     QKV +1.2 %, fc1 +2 %, fp32-residual outputs -7 % (only 4 waves doing the read-modify-write), whole iteration +0.2 %.
// Where the time goes (tools/gemm_bench.py variants 21-36, tools/epi_cost.py): MFMA-only loop 0.30 ms, LDS-DMA + ds_read +
// barriers without MFMA 0.34, both 0.47 (not DMA latency: never waiting for the DMA gives the same time; a deeper ring
// changes nothing), prologue + one K-tile + epilogue 0.12.  The epilogue cost is the chip-wide HBM burst: it scales with
// the bytes (bf16 0.09, fp32 0.16, fp32 read-modify-write 0.32 ms at this shape = 5.5-6.3 TB/s), a lone workgroup's
// epilogue takes < 2 us against 6 us per tile when all 256 CUs store together.  The vendor BLAS runs the same four shapes
// without any epilogue at 1245-1273 TFLOP/s.
// M rows of 256 x 256 tiles (M may be 0) + tail_rows rows of 64 x 64 tail tiles starting at row M, one grid
// the pacing words of the current device (gemm_bf16_pp_kernel: "round pacing"): [0] = switched off by a timeout, [16 + 2 i] /
// [17 + 2 i] = arrivals / departures of (XCD, round) i; zeroed once, re-armed by the kernel itself.  nullptr: no pacing here
// (PGIBBS_GEMM_RSYNC=0, a device that is not 8 XCDs x 32 CUs, allocation failure).
static unsigned* round_sync_words() {
  static unsigned* words[64];
  static bool tried[64];
  // OFF by default: measured round 5 (profiles/r05_fc2_round_pacing.txt) -- pacing cuts fc2's fabric reads from 3.67 to 2.2 GB per
  // launch (FETCH_SIZE) and the launch gets 1 % SLOWER (756 -> 763 us; ESM-MSA-1b's fc2 +3 %): the re-fetches are served by the
  // Infinity Cache at no cost in time, while waiting for the slowest CU of a round is not free.  Kept as the ablation switch.
  static const int on = [] { const char* e = getenv("PGIBBS_GEMM_RSYNC"); return e ? atoi(e) : 0; }();
  int dev = 0;
  if (!on || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!tried[dev]) {
    tried[dev] = true;
    hipDeviceProp_t p;
    void* w = nullptr;
    const size_t bytes = (size_t)(16 + 2 * 8 * RS_ROUNDS) * 4;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount == 256 && hipMalloc(&w, bytes) == hipSuccess) {
      if (hipMemset(w, 0, bytes) == hipSuccess) words[dev] = (unsigned*)w;
    }
  }
  return words[dev];
}

static int launch_pp(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K,
                     int ldx, int ldw, int ldo, int epi, int abl = 0, int tail_rows = 0, int tile_rows = 256) {
  const int tiles_m = M / tile_rows, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  static const int tail_last = [] { const char* e = getenv("PGIBBS_GEMM_TAIL_LAST"); return e ? atoi(e) : 0; }();
  static const int rsync_k = [] { const char* e = getenv("PGIBBS_GEMM_RSYNC_K"); return e ? atoi(e) : 4096; }();
  const int n_tail_abs = (tail_rows / 64) * (N / 64), tail_m0 = M;
  // round pacing: deep-K residual GEMMs of more than two rounds of 256-row tiles (fc2 of a big batch); the tail tiles then ride
  // BEHIND the big ones (in front they would hold the first round's CUs back, and with them everybody who waits for that round)
  unsigned* rsync = (epi == EPI_F32_RESID && !abl && tile_rows == 256 && K >= rsync_k && n_tiles > 2 * 256 && n_tail_abs % 8 == 0)
                        ? round_sync_words() : nullptr;
  const int n_tail = (tail_last || rsync) ? -n_tail_abs : n_tail_abs;
  dim3 grid(n_tiles + n_tail_abs), block(512);
  if (!abl) {
    note_kernel(tile_rows == 192 ? "pp192x256" : "pp256x256", n_tiles);
    if (n_tail_abs) note_kernel("tail64", n_tail_abs);
  }
#define PG_PP_ARGS X, W, bias, out, K, ldx, ldw, ldo, tiles_n, n_tiles, n_tail, tail_m0, rsync
  if (abl) {   // ablations: EPI_BF16 only
    if (abl == 1) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 1>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 2) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 2>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 3) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 3>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 13) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 13>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 14) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_F32_RESID, 13>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 18) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 18>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 19) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 19>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 28) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_F32_RESID, 18>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 29) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_F32_RESID, 0>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 17) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 17>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 16) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 16>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 15) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 15>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 12) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 12>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 10) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 10>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 11) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 11>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 8) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 8>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 9) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 9>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 4) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 0, 1>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 5) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 0, 8>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 6) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 0, 2>), grid, block, 0, s, PG_PP_ARGS);
    if (abl == 7) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_BF16, 0, 16>), grid, block, 0, s, PG_PP_ARGS);
    PG_HIP(hipGetLastError());
    return 0;
  }
  // m-panels per rasterisation group: 4, but 2 for deep K (fc2: a 256-row X panel of K = 5120 is 2.6 MB, four of them plus
  // the W panels overflow the XCD's 4 MB L2 even slice-wise; measured 0.760 -> 0.733 ms with the bf16 epilogue)
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  const int gm = gm_env ? gm_env : (K >= 4096 ? 2 : 4);
  if (tile_rows == 192) {      // 192 x 256 tiles: residual epilogue only (launch_gemm_big)
    if (epi != EPI_F32_RESID || M % 192) return fail(1, "gemm: 192-row tiles are for the residual epilogue");
    if (gm == 2) hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_F32_RESID, 0, 2, 6>), grid, block, 0, s, PG_PP_ARGS);
    else hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI_F32_RESID, 0, 4, 6>), grid, block, 0, s, PG_PP_ARGS);
    PG_HIP(hipGetLastError());
    return 0;
  }
#define PG_GEMM_CASE(E)                                                                                                   \
  case E:                                                                                                                 \
    if (gm == 1) hipLaunchKernelGGL((gemm_bf16_pp_kernel<E, 0, 1>), grid, block, 0, s, PG_PP_ARGS); \
    else if (gm == 2) hipLaunchKernelGGL((gemm_bf16_pp_kernel<E, 0, 2>), grid, block, 0, s, PG_PP_ARGS); \
    else hipLaunchKernelGGL((gemm_bf16_pp_kernel<E>), grid, block, 0, s, PG_PP_ARGS); \
    break;
  switch (epi) {
    PG_GEMM_CASE(EPI_BF16)
    PG_GEMM_CASE(EPI_BF16_GELU)
    PG_GEMM_CASE(EPI_F32_RESID)
    PG_GEMM_CASE(EPI_F32)
    PG_GEMM_CASE(EPI_F32_GELU)
    default:
      return fail(1, "gemm: bad epilogue for the 8-wave tile kernel");
  }
#undef PG_GEMM_CASE
#undef PG_PP_ARGS
  PG_HIP(hipGetLastError());
  return 0;
}

// geometry of a big-batch launch: m-panels of 256 x 256 tiles + rows of 64 x 64 tail tiles (launch_gemm_big)
struct BigGeom { int m_main, tail_rows, gm; };
static BigGeom big_geometry(int M, int N, int K) {
  static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  static const int tail_on = [] { const char* e = getenv("PGIBBS_GEMM_TAIL"); return e ? atoi(e) : 1; }();
  static const int tail_max = [] { const char* e = getenv("PGIBBS_GEMM_TAIL_MAX"); return e ? atoi(e) : 4; }();   // tail tiles per CU at most
  static const int gm_env = [] { const char* e = getenv("PGIBBS_GEMM_GM"); return e ? atoi(e) : 0; }();
  const int tiles_n = N / 256, tiles_m = M / 256;
  const long t256 = (long)tiles_m * tiles_n, full = t256 / n_cu, frac = t256 - full * n_cu;
  BigGeom g{tiles_m, 0, gm_env ? (gm_env <= 2 ? gm_env : 4) : (K >= 4096 ? 2 : 4)};      // the GM the launchers instantiate
  if (tail_on && frac > 0) {
    const int mm = (int)(full * n_cu / tiles_n);             // m-panels that fill whole rounds (0: less than one round of big tiles)
    const long n_tail = (long)(tiles_m - mm) * 4 * (N / 64);
    if (n_tail <= (long)tail_max * n_cu) { g.m_main = mm; g.tail_rows = (tiles_m - mm) * 256; }
  }
  return g;
}
void gemm_big_geometry(int M, int N, int K, int* m_main_panels, int* tail_rows) {
  const BigGeom g = big_geometry(M, N, K);
  *m_main_panels = g.m_main;
  *tail_rows = g.tail_rows;
}
// The big-batch GEMM: whole rounds of 256 x 256 tiles (one per CU) plus, in the SAME grid, 64 x 64 tail tiles for the rows
// beyond the last full round when that round would be mostly empty (gemm_epilogue.h).  Kernel per epilogue: the 16-wave kernel
// for the bf16 outputs (QKV projections: 3.5-4 % faster there; fc1: its one-pass GELU epilogue is 0.03 ms shorter per launch),
// the 8-wave ping-pong kernel for the fp32 outputs (the 16-wave kernel loses 3-4 % on the residual read-modify-write);
// PGIBBS_GEMM_BIG=pp / w16 forces one kernel for the plain epilogues.  M, N multiples of 256, K a multiple of 64, K >= 128.
int launch_gemm_big(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldx,
                    int ldw, int ldo, int epi, int m_live) {
  if (M % 256 || N % 256 || K % 64 || K < 128 || M < 256) return fail(1, "gemm_big: shape");
  static const int big = [] { const char* e = getenv("PGIBBS_GEMM_BIG"); return !e ? -1 : (e[0] == 'w' ? 16 : 0); }();
  const BigGeom geo = big_geometry(M, N, K);
  const int m_main = geo.m_main, tail_rows = geo.tail_rows;
  const bool bf16out = epi == EPI_BF16 || epi == EPI_BF16_GELU;
  const bool use16 = big == 16 || (big == -1 && bf16out);
  static const int n_cu_l = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  // Round 6: the tile-height ladder (gemm_ladder.hip; heights 160 ... 240 in steps of 16 rows).  Built for VERDICT r05's tile-count
  // argument -- a partial last round of tiles costs a whole one, and N = 1280 (five tiles per row panel) quantises badly for a
  // 1/8 ... 1/2 shard of config 3 -- and MEASURED (profiles/r06_ladder_bench.txt, r06_shard_proxy_ladder_ab.txt): every height
  // takes the same time to within 3 % at every shard size (a 32-chain shard's fc2: 165 tiles of 256 rows 128.8 us, 220 of 192
  // 127.0, 240 of 176 133.6, 190 of 224 128.7), bit-identical results.  These launches run at a fixed aggregate L2 -> LDS feed of
  // ~8 TB/s whatever the tiling (bytes staged / time: 6.8 ... 8.5 TB/s from the 32-chain shard to the full batch), so idle CUs in a
  // partial round are not lost throughput and a smaller tile only raises the bytes staged per FLOP.  OFF by default;
  // PGIBBS_GEMM_LADDER=1 prices the candidates in rounds x height (the model that does not hold), =h forces height h where the
  // epilogue has it (the A/B switch of tests/test_gpu_kernels.py and tools/ladder_bench.py).
  static const int ladder = [] { const char* e = getenv("PGIBBS_GEMM_LADDER"); return e ? atoi(e) : 0; }();
  const int live = m_live > 0 && m_live <= M ? m_live : M;
  const int tiles_n_l = N / 256;
  double cost_old;
  {
    const long t256 = (long)(M / 256) * tiles_n_l;
    cost_old = tail_rows ? (double)((long)m_main * tiles_n_l / n_cu_l) + 0.15 : (double)((t256 + n_cu_l - 1) / n_cu_l);
    static const int t192e = [] { const char* e = getenv("PGIBBS_GEMM_T192"); return e ? atoi(e) : 1; }();
    if (t192e && epi == EPI_F32_RESID) {
      const int m192 = M / 192, rest = M - m192 * 192;
      const double r192 = 0.75 * (double)(((long)m192 * tiles_n_l + n_cu_l - 1) / n_cu_l) + (rest ? 0.2 : 0.0);
      if (r192 <= 0.9 * cost_old) cost_old = r192;
    }
  }
  if (ladder && (epi == EPI_F32_RESID || epi == EPI_BF16_GELU)) {
    int best_h = 0;
    double best = 0.95 * cost_old;
    for (int h = 240; h >= 160; h -= 16) {
      if (!gemm_ladder_has(epi, h) || (ladder > 1 && ladder != h)) continue;
      const long tiles = (long)((live + h - 1) / h) * tiles_n_l;
      const double c = (double)((tiles + n_cu_l - 1) / n_cu_l) * h / 256.0;
      if (ladder == h || c < best - 1e-9) { best = c; best_h = h; if (ladder == h) break; }
    }
    if (best_h) return launch_gemm_ladder(s, X, W, bias, out, live, M, best_h, N, K, ldx, ldw, ldo, epi);
  }
  if (use16) return launch_gemm_w16(s, X, W, bias, out, m_main * 256, N, K, ldx, ldw, ldo, epi, 0, tail_rows);
  // Residual GEMMs of mid-size batches: 256-row tiles quantise badly (a 32-chain shard's out-projection / fc2: 165 tiles on 256 CUs,
  // a 64-chain one: 325 = two rounds of which the second is a quarter full).  192 x 256 tiles (same kernel, 6 instead of 8 row
  // blocks per wave, same k order) cost three quarters of a round each: taken when that is at least 10 % fewer round-equivalents.
  // The rows beyond the last 192-row tile (0, 64 or 128 of them: M is a multiple of 256) are 64 x 64 tail tiles of the same grid.
  static const int t192 = [] { const char* e = getenv("PGIBBS_GEMM_T192"); return e ? atoi(e) : 1; }();
  if (t192 && epi == EPI_F32_RESID) {
    static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
    const int tiles_n = N / 256;
    const long t256 = (long)(M / 256) * tiles_n;
    const double r256 = tail_rows ? (double)((long)m_main * tiles_n / n_cu) + 0.15 : (double)((t256 + n_cu - 1) / n_cu);
    const int m192 = M / 192, rest = M - m192 * 192;
    const long t192n = (long)m192 * tiles_n;
    // leftover rows cost a round of tail tiles in front of the big ones (~0.2 of a round: M = 9728, N = 1280 measured 51.2 us with
    // 250 tiles of 192 rows + 40 tail tiles against 47.7 us for 190 tiles of 256 rows; tools/gemm_mid_resid_bench.py)
    const double r192 = 0.75 * (double)((t192n + n_cu - 1) / n_cu) + (rest ? 0.2 : 0.0);
    if (t192 == 2 || r192 <= 0.9 * r256) return launch_pp(s, X, W, bias, out, m192 * 192, N, K, ldx, ldw, ldo, epi, 0, rest, 192);
  }
  return launch_pp(s, X, W, bias, out, m_main * 256, N, K, ldx, ldw, ldo, epi, 0, tail_rows);
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM for M <= 256 token rows (BASELINE config 1: one chain of 27 tokens; small interactive jobs; LM-head GEMMs).
// With so few rows the op is pure weight streaming (HBM roofline: 1.3 GB of bf16 weights per forward), and a 256x256
// tile grid would put 5-20 workgroups on 256 CUs.  Here: one workgroup per 16 output features (N/16 = 80..320
// workgroups), its 4 waves split K four ways; W fragments go global -> VGPR directly (each byte is used once, an LDS
// round trip would be pure overhead), the tiny X operand comes from L2; partial sums meet in LDS.
// ------------------------------------------------------------------------------------------------
template <int MT, int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_bf16_skinny_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out, int K,
                                                              int ldx, int ldw, int ldo, long split_stride) {
  if (EPI == EPI_F32_PARTIAL) {
    X += (size_t)blockIdx.y * K;
    W += (size_t)blockIdx.y * K;
    out = (float*)out + (size_t)blockIdx.y * split_stride;
  }
  __shared__ float red[NW - 1][MT][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int kq = K / NW;                                 // this wave's K range: a multiple of 16 (NW=4) or 32 (NW=8)
  const bf16_t* wp = W + (size_t)(n0 + fr) * ldw + wave * kq + fq * 8;
  const bf16_t* xp = X + (size_t)fr * ldx + wave * kq + fq * 8;
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int k = 0;
  // MFMA k-steps per trip (all loads issued first).  5 for up to two m-tiles: a wave's share of K = 1280 with 8 waves is 160 =
  // 5 x 32, ONE batch of loads instead of 4 + 1 -- the kernel is a chain of memory round trips (same k order, same bits)
  constexpr int U = MT <= 2 ? 5 : (MT <= 4 ? 4 : (MT <= 8 ? 2 : 1));
  for (; k + 32 * U <= kq; k += 32 * U) {
    bf16x8 wf[U], xf[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      wf[u] = *(const bf16x8*)(wp + k + u * 32);
#pragma unroll
      for (int t = 0; t < MT; ++t) xf[u][t] = *(const bf16x8*)(xp + (size_t)t * 16 * ldx + k + u * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = mfma_op16(wf[u], xf[u][t], acc[t]);
  }
  for (; k < kq; k += 32) {
    if (k + 32 <= kq) {
      const bf16x8 wf = *(const bf16x8*)(wp + k);
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = mfma_op16(wf, *(const bf16x8*)(xp + (size_t)t * 16 * ldx + k), acc[t]);
    } else {                                             // 16-wide tail of the wave's range: upper k-chunks are zero
      bf16x8 wf, xz;
#pragma unroll
      for (int e = 0; e < 8; ++e) { wf[e] = (__bf16)0.f; xz[e] = (__bf16)0.f; }
      if (fq < 2) wf = *(const bf16x8*)(wp + k);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        bf16x8 xf = xz;
        if (fq < 2) xf = *(const bf16x8*)(xp + (size_t)t * 16 * ldx + k);
        acc[t] = mfma_op16(wf, xf, acc[t]);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)&red[wave - 1][t][lane][0] = acc[t];
  }
  __syncthreads();
  if (wave > 0) return;
  // lane holds D[n = n0 + fq*4 + r][m = t*16 + fr]
  const float4 b4 = EPI == EPI_F32_PARTIAL ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(bias + n0 + fq * 4);
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    f32x4 v = acc[t];
#pragma unroll
    for (int w2 = 0; w2 < NW - 1; ++w2) {
      const f32x4 o = *(const f32x4*)&red[w2][t][lane][0];
      v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
    }
    float v0 = v[0] + b4.x, v1 = v[1] + b4.y, v2 = v[2] + b4.z, v3 = v[3] + b4.w;
    if (EPI == EPI_F32_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
    const size_t o = (size_t)(t * 16 + fr) * ldo + n0 + fq * 4;
    if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
      uint2 p;
      p.x = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v0, v1) : pack_op2(v0, v1);
      p.y = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v2, v3) : pack_op2(v2, v3);
      *(uint2*)((bf16_t*)out + o) = p;
    } else if (EPI == EPI_F32_RESID) {
      float4* dst = (float4*)((float*)out + o);
      float4 r = *dst;
      r.x += v0; r.y += v1; r.z += v2; r.w += v3;
      *dst = r;
    } else {
      *(float4*)((float*)out + o) = make_float4(v0, v1, v2, v3);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same weight-streaming GEMM with the preceding LayerNorm folded into its operand load (single chains: BASELINE config 1).
//   out = LayerNorm(x; gamma, beta) . W^T + bias,   x fp32 [M <= 32 rows][K = d_model]
// In the launch-bound regime a LayerNorm is a whole kernel (>= 4 us on this part, whatever it computes) for 140 KB of data
// that the next GEMM's workgroups read anyway: every workgroup here loads the fp32 residual rows itself -- wave w its K slice
// of every row --, the eight waves combine the row sums through LDS (two-pass: mean, then the centred sum of squares, on the
// values held in registers), and each wave normalises its slice on the way into the MFMA.  No extra memory traffic (the
// bf16 operand rows were re-read by every workgroup before; now it is the fp32 rows, from L2), two kernels per layer less.
// The statistics are summed in another order than ln_row.h's (few-chain regime: kernels are picked by the local shape).
// ------------------------------------------------------------------------------------------------
// NB = 16-feature blocks per workgroup: 1, or 2 when N / 16 workgroups would not be resident at once (the row data is 80 VGPRs:
// one 8-wave workgroup per CU) -- fc1's 320 then run as 160 workgroups in one dispatch round instead of 256 + 64.
template <int MT, int EPI, int NKS, int NB>
__global__ __launch_bounds__(512) void gemm_ln_skinny_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                           void* __restrict__ out, int K, int ldw, int ldo) {
  constexpr int NW = 8;
  __shared__ float red[NW - 1][MT][NB][64][4];
  __shared__ float stat[2][NW][MT][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int n0 = blockIdx.x * 16 * NB;
  const int kq = NKS * 32;                               // this wave's K range (K = 8 * kq)
  const int k0 = wave * kq + fq * 8;
  const bf16_t* wp = W + (size_t)(n0 + fr) * ldw + k0;
  const float* xp = X + (size_t)fr * ldx + k0;
  bf16x8 wf[NKS][NB];
  float4 xa[NKS][MT][2];
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wf[u][nb] = *(const bf16x8*)(wp + (size_t)nb * 16 * ldw + u * 32);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      xa[u][t][0] = *(const float4*)(xp + (size_t)t * 16 * ldx + u * 32);
      xa[u][t][1] = *(const float4*)(xp + (size_t)t * 16 * ldx + u * 32 + 4);
    }
  }
  const float inv_k = 1.0f / (float)K;
  float mean[MT], rstd[MT];
  // pass 1: row means (row = t*16 + fr; the four fq lanes of a row, then the eight waves)
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NKS; ++u)
      s += ((xa[u][t][0].x + xa[u][t][0].y) + (xa[u][t][0].z + xa[u][t][0].w)) + ((xa[u][t][1].x + xa[u][t][1].y) + (xa[u][t][1].z + xa[u][t][1].w));
    s = rows4_sum(s);
    if (fq == 0) stat[0][wave][t][fr] = s;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) s += stat[0][w2][t][fr];
    mean[t] = s * inv_k;
  }
  // pass 2: centred sums of squares
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NKS; ++u)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4& v = xa[u][t][hh];
        v.x -= mean[t]; v.y -= mean[t]; v.z -= mean[t]; v.w -= mean[t];
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
    q = rows4_sum(q);
    if (fq == 0) stat[1][wave][t][fr] = q;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float q = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) q += stat[1][w2][t][fr];
    rstd[t] = 1.0f / sqrtf(q * inv_k + eps);
  }
  f32x4 acc[MT][NB];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
    const float4 g0 = *(const float4*)(gamma + k0 + u * 32), g1 = *(const float4*)(gamma + k0 + u * 32 + 4);
    const float4 b0 = *(const float4*)(beta + k0 + u * 32), b1 = *(const float4*)(beta + k0 + u * 32 + 4);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float4 a = xa[u][t][0], b = xa[u][t][1];
      const float r = rstd[t];
      uint4 pk;
      pk.x = pack_op2(a.x * r * g0.x + b0.x, a.y * r * g0.y + b0.y);
      pk.y = pack_op2(a.z * r * g0.z + b0.z, a.w * r * g0.w + b0.w);
      pk.z = pack_op2(b.x * r * g1.x + b1.x, b.y * r * g1.y + b1.y);
      pk.w = pack_op2(b.z * r * g1.z + b1.z, b.w * r * g1.w + b1.w);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[t][nb] = mfma_op16(wf[u][nb], __builtin_bit_cast(bf16x8, pk), acc[t][nb]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) *(f32x4*)&red[wave - 1][t][nb][lane][0] = acc[t][nb];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const float4 b4 = *(const float4*)(bias + n0 + nb * 16 + fq * 4);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f32x4 v = acc[t][nb];
#pragma unroll
      for (int w2 = 0; w2 < NW - 1; ++w2) {
        const f32x4 o = *(const f32x4*)&red[w2][t][nb][lane][0];
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
      }
      const float v0 = v[0] + b4.x, v1 = v[1] + b4.y, v2 = v[2] + b4.z, v3 = v[3] + b4.w;
      uint2 p;
      p.x = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v0, v1) : pack_op2(v0, v1);
      p.y = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v2, v3) : pack_op2(v2, v3);
      *(uint2*)((bf16_t*)out + (size_t)(t * 16 + fr) * ldo + n0 + nb * 16 + fq * 4) = p;
    }
  }
}

// may the LayerNorm-folding weight-streaming GEMM take this shape?  (M rows incl. padding; K = the normalised width)
bool gemm_ln_skinny_ok(int M, int N, int K) {
  static const int on = [] { const char* e = getenv("PGIBBS_LN_SKINNY"); return e ? atoi(e) : 1; }();
  return on && (M == 16 || M == 32) && N % 16 == 0 && K % 256 == 0 && K / 256 >= 1 && K / 256 <= 5;
}
int launch_gemm_ln_skinny(hipStream_t s, const float* X, int ldx, const float* gamma, const float* beta, float eps, const bf16_t* W,
                          const float* bias, void* out, int M, int N, int K, int ldw, int ldo, int epi) {
  if (!gemm_ln_skinny_ok(M, N, K) || (epi != EPI_BF16 && epi != EPI_BF16_GELU)) return fail(1, "gemm_ln_skinny: shape / epilogue");
  const int nks = K / 256;
  static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  static const int nb_env = [] { const char* e = getenv("PGIBBS_LN_SKINNY_NB"); return e ? atoi(e) : 0; }();
  const int nb = nb_env ? nb_env : ((N / 16 > n_cu && N % 32 == 0) ? 2 : 1);
  if (nb == 2 && N % 32) return fail(1, "gemm_ln_skinny: N must be a multiple of 32 for two feature blocks per workgroup");
  dim3 grid(N / (16 * nb)), block(512);
  note_kernel("ln+skinny8w", N / (16 * nb));
#define PG_LNS(MTV, E, NK)                                                                                                       \
  do {                                                                                                                           \
    if (nb == 2) hipLaunchKernelGGL((gemm_ln_skinny_kernel<MTV, E, NK, 2>), grid, block, 0, s, X, ldx, gamma, beta, eps, W, bias, out, K, ldw, ldo); \
    else hipLaunchKernelGGL((gemm_ln_skinny_kernel<MTV, E, NK, 1>), grid, block, 0, s, X, ldx, gamma, beta, eps, W, bias, out, K, ldw, ldo);     \
  } while (0)
#define PG_LNS_NK(MTV, E)                                                       \
  switch (nks) {                                                                \
    case 1: PG_LNS(MTV, E, 1); break;                                           \
    case 2: PG_LNS(MTV, E, 2); break;                                           \
    case 3: PG_LNS(MTV, E, 3); break;                                           \
    case 4: PG_LNS(MTV, E, 4); break;                                           \
    default: PG_LNS(MTV, E, 5); break;                                          \
  }
  if (M == 16) {
    if (epi == EPI_BF16) { PG_LNS_NK(1, EPI_BF16) } else { PG_LNS_NK(1, EPI_BF16_GELU) }
  } else {
    if (epi == EPI_BF16) { PG_LNS_NK(2, EPI_BF16) } else { PG_LNS_NK(2, EPI_BF16_GELU) }
  }
#undef PG_LNS_NK
#undef PG_LNS
  PG_HIP(hipGetLastError());
  return 0;
}

template <int MT>
static int launch_skinny_mt(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int N, int K, int ldx,
                            int ldw, int ldo, int epi, int splits = 1, long split_stride = 0) {
  // 8 waves per workgroup (K split 8 ways: twice the loads in flight per CU) whenever each wave still gets whole
  // 32-wide k-steps and the m-tile count keeps the register footprint small
  const bool w8 = (K % 256 == 0) && MT <= 4;
  dim3 grid(N / 16, splits), block(w8 ? 512 : 256);
  note_kernel(w8 ? "skinny8w" : "skinny4w", N / 16, splits);
#define PG_GEMM_CASE(E)                                                                                              \
  case E:                                                                                                            \
    if (w8) hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, E, 8>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, split_stride); \
    else hipLaunchKernelGGL((gemm_bf16_skinny_kernel<MT, E, 4>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, split_stride);    \
    break;
  switch (epi) {
    PG_GEMM_CASE(EPI_BF16)
    PG_GEMM_CASE(EPI_BF16_GELU)
    PG_GEMM_CASE(EPI_F32_RESID)
    PG_GEMM_CASE(EPI_F32)
    PG_GEMM_CASE(EPI_F32_GELU)
    PG_GEMM_CASE(EPI_F32_PARTIAL)
    default:
      return fail(1, "gemm: bad epilogue");
  }
#undef PG_GEMM_CASE
  PG_HIP(hipGetLastError());
  return 0;
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N,
                      int K, int ldx, int ldw, int ldo, int epi, int splits = 1, long split_stride = 0) {
  const int tiles_m = M / BM, tiles_n = N / BN, n_tiles = tiles_m * tiles_n;
  dim3 grid(n_tiles, splits), block((BM / WM) * (BN / WN) * 64);
  note_kernel(BM == 64 ? "tile64x64" : (BM == 128 ? "tile128x128" : "tile256x256-lockstep"), n_tiles, splits);
#define PG_GEMM_CASE(E)                                                                                              \
  case E:                                                                                                            \
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, E>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, \
                       tiles_n, n_tiles, split_stride);                                                     \
    break;
  switch (epi) {
    PG_GEMM_CASE(EPI_BF16)
    PG_GEMM_CASE(EPI_BF16_GELU)
    PG_GEMM_CASE(EPI_F32_RESID)
    PG_GEMM_CASE(EPI_F32)
    PG_GEMM_CASE(EPI_F32_GELU)
    PG_GEMM_CASE(EPI_F32_PARTIAL)
    default:
      return fail(1, "gemm: bad epilogue");
  }
#undef PG_GEMM_CASE
  PG_HIP(hipGetLastError());
  return 0;
}

// out[m][n] += bias[n] + sum over the K-splits, in split order (fixed -> bit-reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                           float* __restrict__ out, int n4, int ldo, long total4, int splits,
                                                           long split_stride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int m = (int)(i / n4), c = (int)(i - (long)m * n4);
  float4 a = *(const float4*)(bias + c * 4);
  for (int sidx = 0; sidx < splits; ++sidx) {
    const float4 p = *(const float4*)(part + (size_t)sidx * split_stride + (size_t)i * 4);
    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
  }
  float4* o = (float4*)(out + (size_t)m * ldo + c * 4);
  float4 r = *o;
  r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
  *o = r;
}

int launch_gemm_bf16(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K,
                     int ldx, int ldw, int ldo, int epi, float* ws, size_t ws_bytes, int m_live) {
  static const int variant = [] { const char* e = getenv("PGIBBS_GEMM"); return e ? atoi(e) : 2; }();
  return launch_gemm_bf16_variant(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi, variant, ws, ws_bytes, m_live);
}

#ifndef PG_F16
// Strict-mode projection on split operands (X3 [M][3K], W3 [N][3K]; K = logical depth): the fused three-product kernel when its
// 256 x 256 tiles fill the chip, else the plain GEMM over K' = 3K -- same summation order, bit-identical results
// (PGIBBS_SPLIT3_FUSED=0 forces the plain form; tests compare the two).
// Will launch_gemm_split3 take the fused kernel for this shape?  The fused kernel (and its tail tiles) read only the [lo | hi]
// blocks of an activation row's groups, the plain kernels all three: a producer may leave the duplicate hi block unwritten
// exactly when this says yes for its consumer (Engine: LayerNorm rows, fc1's EPI_SPLIT2_GELU rows).
bool gemm_split3_fused(int M, int N, int K, int epi) {
  static const int fused = [] { const char* e = getenv("PGIBBS_SPLIT3_FUSED"); return e ? atoi(e) : 1; }();
  const bool ok256 = M % 256 == 0 && N % 256 == 0 && K % 32 == 0 && M >= 256;
  return ok256 && (epi == EPI_SPLIT3_GELU || epi == EPI_SPLIT2_GELU || (fused && (long)(M / 256) * (N / 256) >= 128));
}
int launch_gemm_split3(hipStream_t s, const bf16_t* X3, const bf16_t* W3, const float* bias, void* out, int M, int N, int K, int ldo,
                       int epi) {
  if (gemm_split3_fused(M, N, K, epi)) return launch_gemm_split3_w16(s, X3, W3, bias, out, M, N, K, ldo, epi);
  if (epi == EPI_SPLIT3_GELU || epi == EPI_SPLIT2_GELU) return fail(1, "gemm: split-operand epilogue needs M, N multiples of 256");
  return launch_gemm_bf16(s, X3, W3, bias, out, M, N, 3 * K, 3 * K, 3 * K, ldo, epi);
}
#endif  // !PG_F16

int launch_gemm_bf16_variant(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N,
                             int K, int ldx, int ldw, int ldo, int epi, int variant, float* ws, size_t ws_bytes, int m_live) {
  // Dispatch by how many tiles each kernel would put on the 256 CUs (times in us, tools/gemm_mid_bench.py, N=1280 K=1280):
  //   M =    16    64   256  1024  4096  16384
  //   skinny 4.3   8.7  26.4                       one workgroup per 16 output features, weights streamed once
  //   64^2         7.7   7.9  13.6  31.7           5 workgroups per CU: fills the chip from 64 token rows on
  //   128^2             16.4  19.8  26.9   78
  //   256^2 ping-pong         31.4  36.4   87      (wins from >= 128 tiles: 41 vs 49 us at M = 8192)
  if (K % 64) return fail(1, "gemm: K must be a multiple of 64");
  // Deep K with few tiles (fc2 of a few dozen chains: K = 5120, 80-320 tiles): K-splits run side by side into ws, then one
  // reduction adds them to the residual stream in fixed order.  M = 1024: 43 -> 21 us, M = 32: 15.6 -> 8 us.
  if (epi == EPI_F32_RESID && ws && K >= 2048 && M >= 16 && M % 16 == 0 && N % 64 == 0 && variant != 1 && variant < 6) {
    const int splits = K % 1280 == 0 ? K / 1280 : (K % 1024 == 0 ? K / 1024 : 1);
    const int Mr = M <= 16 ? 16 : (M <= 32 ? 32 : (M <= 256 ? (M + 63) / 64 * 64 : M));   // rows the kernels write (their tile heights)
    const long t64 = (long)(Mr / 64) * (N / 64);
    const bool pp_sized = M % 256 == 0 && N % 256 == 0 && (long)(M / 256) * (N / 256) >= 128;   // the 256^2 kernel fills the chip
    const bool small = M <= 48 || ((M <= 256 || M % 64 == 0) && t64 * splits <= 1280 && !pp_sized);
    const long stride = (long)Mr * N;
    // (round 4: ONE launch of N / 16 workgroups with 16 waves splitting K instead of the K-splits + reduction was measured slower,
    // config 1 1.73 -> 1.92 ms per iteration: every workgroup re-reads the whole [M][K] operand, 327 KB at K = 5120, and a CU's
    // L2 -> L1 path carries ~130 GB/s -- the K-splits spread those reads over four times as many CUs)
    if (splits > 1 && small && (size_t)splits * stride * 4 <= ws_bytes && (K / splits) % 64 == 0) {
      int rc;
      const int Ks = K / splits;
      if (M <= 48) {
        const int mt = M / 16;
        rc = mt <= 1 ? launch_skinny_mt<1>(s, X, W, bias, ws, N, Ks, ldx, ldw, N, EPI_F32_PARTIAL, splits, stride)
           : mt <= 2 ? launch_skinny_mt<2>(s, X, W, bias, ws, N, Ks, ldx, ldw, N, EPI_F32_PARTIAL, splits, stride)
                     : launch_skinny_mt<4>(s, X, W, bias, ws, N, Ks, ldx, ldw, N, EPI_F32_PARTIAL, splits, stride);
      } else if (Mr % 128 == 0 && N % 128 == 0 && (long)(Mr / 128) * (N / 128) * splits >= 256) {
        rc = launch_cfg<128, 128, 64, 64>(s, X, W, bias, ws, Mr, N, Ks, ldx, ldw, N, EPI_F32_PARTIAL, splits, stride);   // enough 128^2 tiles
      } else {
        rc = launch_cfg<64, 64, 32, 32>(s, X, W, bias, ws, Mr, N, Ks, ldx, ldw, N, EPI_F32_PARTIAL, splits, stride);
      }
      if (rc) return rc;
      const long total4 = stride / 4;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ws, bias, (float*)out, N / 4,
                         ldo, total4, splits, stride);
      PG_HIP(hipGetLastError());
      return 0;
    }
  }
  if (variant == 6 && M % 64 == 0 && N % 64 == 0) return launch_cfg<64, 64, 32, 32>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
  if (variant == 7 && M % 128 == 0 && N % 128 == 0) return launch_cfg<128, 128, 64, 64>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
  if (variant == 8 && epi == EPI_F32_RESID && M % 256 == 0 && N % 256 == 0 && K >= 128 && M >= 192)      // micro-benchmark: 192-row tiles
    return launch_pp(s, X, W, bias, out, (M / 192) * 192, N, K, ldx, ldw, ldo, epi, 0, M - (M / 192) * 192, 192);
  if (M >= 16 && M <= 48 && M % 16 == 0 && N % 16 == 0 && variant != 1) {
    const int mt = M / 16;
    if (mt <= 1) return launch_skinny_mt<1>(s, X, W, bias, out, N, K, ldx, ldw, ldo, epi);
    if (mt <= 2) return launch_skinny_mt<2>(s, X, W, bias, out, N, K, ldx, ldw, ldo, epi);
    return launch_skinny_mt<4>(s, X, W, bias, out, N, K, ldx, ldw, ldo, epi);
  }
  if (M <= 256) {
    // rows beyond M inside the last 64-row tile are read/written too: callers pad buffers to 256 rows
    if (M % 16 || N % 64) return fail(1, "gemm: M must be a multiple of 16 and N of 64");
    return launch_cfg<64, 64, 32, 32>(s, X, W, bias, out, (M + 63) / 64 * 64, N, K, ldx, ldw, ldo, epi);
  }
  if (M % 128 || N % 64) return fail(1, "gemm: M must be a multiple of 128 and N of 64");
  const bool ok256 = M % 256 == 0 && N % 256 == 0 && K >= 128, ok128 = N % 128 == 0;
  const long t256 = (long)(M / 256) * (N / 256), t128 = (long)(M / 128) * (N / 128);
  if (variant >= 80 && ok256) return launch_gemm_w16(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi, variant - 80);
  if (variant >= 60 && ok256) return launch_pp(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi, variant - 40);   // pp ablations 20..
  if (variant >= 20 && ok256) return launch_pp(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi, variant - 20);
  if (variant == 1) {
    if (ok256) return launch_cfg<256, 256, 128, 64>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
    if (ok128) return launch_cfg<128, 128, 64, 64>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
  }
  if (ok256 && t256 >= 128) return launch_gemm_big(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi, m_live);
  if (ok128 && (t128 >= 200 || !(M % 64 == 0))) return launch_cfg<128, 128, 64, 64>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
  return launch_cfg<64, 64, 32, 32>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
}

PG_OPS_END
