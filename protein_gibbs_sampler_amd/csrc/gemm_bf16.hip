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
//     for one token m -> 8-byte bf16 / 16-byte fp32 row-major stores, no transpose.
//   * 256x256x64 block tile, 8 waves (2 over m x 4 over n), wave tile 128(m) x 64(n),
//     v_mfma_f32_16x16x32_bf16; 128x128 variant (4 waves) for narrow shapes.
//   * HBM/L2 -> LDS by direct `global_load_lds` (16 B/lane, no VGPR round trip), double-buffered;
//     LDS rows are 128 B (64 k) and 16-B chunks are XOR-swizzled with (row & 7) by permuting the
//     per-lane SOURCE address (the LDS image of a glds is lane-linear), so every ds_read_b128
//     lane group touches 16 distinct 16-B slots (conflict-free).
//   * 1-D grid remapped so each XCD (private L2) owns a contiguous range of tiles.
//   * epilogues fused: +bias, erf-GELU, bf16 pack, fp32 residual read-modify-write.
#include "kernels.h"

namespace pg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PG_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

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
    int K, int ldx, int ldw, int ldo, int tiles_n, int n_tiles) {
  constexpr int NWM = BM / WM, NWN = BN / WN, NW = NWM * NWN;
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane holds D[n = nb + fq*4 + r][m = mb + fr], r = 0..3
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * WN + i * 16 + fq * 4;
    const float4 b4 = *(const float4*)(bias + n);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * WM + j * 16 + fr;
      float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
      if (EPI == EPI_BF16_GELU || EPI == EPI_F32_GELU) {
        v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
      }
      if (EPI == EPI_BF16 || EPI == EPI_BF16_GELU) {
        uint2 p;
        p.x = pack_bf16x2(v0, v1);
        p.y = pack_bf16x2(v2, v3);
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

template <int BM, int BN, int WM, int WN>
static int launch_cfg(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N,
                      int K, int ldx, int ldw, int ldo, int epi) {
  const int tiles_m = M / BM, tiles_n = N / BN, n_tiles = tiles_m * tiles_n;
  dim3 grid(n_tiles), block((BM / WM) * (BN / WN) * 64);
#define PG_GEMM_CASE(E)                                                                                              \
  case E:                                                                                                            \
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, E>), grid, block, 0, s, X, W, bias, out, K, ldx, ldw, ldo, \
                       tiles_n, n_tiles);                                                                            \
    break;
  switch (epi) {
    PG_GEMM_CASE(EPI_BF16)
    PG_GEMM_CASE(EPI_BF16_GELU)
    PG_GEMM_CASE(EPI_F32_RESID)
    PG_GEMM_CASE(EPI_F32)
    PG_GEMM_CASE(EPI_F32_GELU)
    default:
      return fail(1, "gemm: bad epilogue");
  }
#undef PG_GEMM_CASE
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_gemm_bf16(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K,
                     int ldx, int ldw, int ldo, int epi) {
  if (M % 128 || N % 128 || K % 64) return fail(1, "gemm: M,N must be multiples of 128 and K of 64");
  if (M % 256 == 0 && N % 256 == 0) return launch_cfg<256, 256, 128, 64>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
  return launch_cfg<128, 128, 64, 64>(s, X, W, bias, out, M, N, K, ldx, ldw, ldo, epi);
}

}  // namespace pg
