// ESM-MSA-1b column attention fused into its QKV projection (round 4):
//   ctx[b, r, c, h*64:(h+1)*64] = softmax_j( q[b,r,c,h] . k[b,j,c,h] ) @ v[b,:,c,h],   [q | k | v] = LN_col(x) W^T + bias
// (fair-esm ColumnSelfAttention behind `self.model.model(batch)["logits"]`, /root/reference/src/pgen/esm_msa_sampler.py:136,236).
//
// The unfused path writes q, k, v of every token (2.4 GB per layer at BASELINE config 4) and reads them back in 197 k two-block
// attention workgroups whose sequences are 32 pieces of 128 B, 1.2 MB apart: 1750 us of GEMM + 701 us of attention per layer.  A
// column's sequence is only R = 32 ... 256 tokens, though, so with the operand rows in COLUMN-MAJOR order (row (b, c, r): the
// preceding LayerNorm writes them there, elementwise.hip) one 256-row GEMM tile holds 256 / R whole sequences, and with the
// weight rows grouped per head ([q_h | k_h | v_h], 192 rows) one 256 x 192 tile holds everything the attention of those sequences
// and that head needs.  The tile's q, k, v go to LDS as bf16 (three 256 x 64 planes in the layout attention.hip uses), each of
// the 16 waves runs the attention of one (sequence, 16-query block) from there, and only the context rows leave the CU -- in the
// ordinary token order, so nothing downstream changes.
//
//   * main loop: the 16-wave kernel's (gemm_w16.hip) with a 256 x 192 tile: wave tile 64 x 48, K-steps of 64 in two 56-KB slots
//     (waves 0-7 stage the 32 activation pieces, waves 8-15 the 24 weight pieces), one s_barrier per K-step; same MFMA, operand
//     roles and k order as every tile kernel, so q, k, v are the very bf16 values the unfused projection stores.
//   * attention: attention_kernel's arithmetic, statement for statement (S^T = K.Q^T, exact softmax in registers, P.V through
//     ds_read_b64_tr_b16) -- the context is bit-identical with the unfused path (tests/test_gpu_msa.py), so fused and unfused
//     launches can be mixed freely across shards and batch sizes.
//   * depth R in {32, 64, 128, 256} (KB = R / 16 key blocks, a template parameter); other depths, <pad> batches and the strict
//     mode take the unfused path.
#include <stdlib.h>

#include "gemm_epilogue.h"

PG_OPS_BEGIN

constexpr int CA_SLOT = 56 * 1024;     // 256 activation rows + 192 weight rows of 128 B

template <int KB, int GM>
__global__ __launch_bounds__(1024, 1) void gemm_colattn_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ ctx, int K,
                                                              int ldx, int ldw, int ld_ctx, int tiles_n, int n_tiles, int R, int C,
                                                              int64_t m_real) {
  __shared__ __attribute__((aligned(16))) char smem[2 * CA_SLOT];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 3, wn = wave >> 2;           // wave tile: activation rows wm*64 .., weight rows wn*48 ..

  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = n_tiles >> 3, r = n_tiles & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tile_m, tile_n;                                // tile_n = head
  {
    const int tiles_m = n_tiles / tiles_n;
    const int gsz = GM * tiles_n, g = bid / gsz, within = bid - g * gsz;
    const int rows = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tile_m = g * GM + within % rows;
    tile_n = within / rows;
  }
  const int m0 = tile_m * 256, n0 = tile_n * 192;

  // ---- LDS-DMA: a slot holds 56 pieces of 1 KiB (0-31: activation rows 8p .., 32-55: weight rows); waves 0-7 stage 4
  // activation pieces each, waves 8-15 three weight pieces each
  const bool stage_w = wave >= 8;
  const int ld_ = stage_w ? ldw : ldx;
  const int npc = stage_w ? 3 : 4;
  const bf16_t* src = stage_w ? W + (size_t)(n0 + (wave - 8) * 24) * ldw : X + (size_t)(m0 + wave * 32) * ldx;
  const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, ((npc * 8 - 1) * ld_ + K) * 2, 0x00020000);
  const int dma_voff = ((lane >> 3) * ld_ + ((lane & 7) ^ (lane >> 3)) * 8) * 2;
  const int piece_bytes = 8 * ld_ * 2;
  const int lds_piece0 = stage_w ? 32 * 1024 + (wave - 8) * 3 * 1024 : wave * 4 * 1024;
  const int nk = K / 64;
  auto dma_step = [&](int t) {
    char* dst = smem + (t & 1) * CA_SLOT + lds_piece0;
    const int soff = t * 128;
#pragma unroll
    for (int g = 0; g < 3; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst + g * 1024), 16, dma_voff, soff + g * piece_bytes, 0, 0);
    if (!stage_w) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, PG_LDS_PTR(dst + 3 * 1024), 16, dma_voff, soff + 3 * piece_bytes, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int foff0 = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const int foff1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
  const int xbase = wm * 4 * 2048;
  const int wbase = 32 * 1024 + wn * 3 * 2048;

  f32x4 acc[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  dma_step(0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of K-step t have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                         // everybody's have; everybody is done reading slot (t+1)&1
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) dma_step(t + 1);
    const char* sb = smem + (t & 1) * CA_SLOT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
      bf16x8 wf[3], xf[4];
#pragma unroll
      for (int i = 0; i < 3; ++i) wf[i] = *(const bf16x8*)(sb + wbase + i * 2048 + fo);
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(sb + xbase + j * 2048 + fo);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_op16(wf[i], xf[j], acc[i][j]);
    }
  }
  __syncthreads();                                        // every wave is done with the operand ring

  // ---- q, k, v of the tile as bf16 into three 256 x 64 planes (rows of 128 B, 16-B chunks XOR-swizzled with row & 7: the K / V
  // tile layout of attention.hip).  acc[i][j][r] = D[n = wn*48 + i*16 + fq*4 + r][m = wm*64 + j*16 + fr]
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int n_loc = wn * 48 + i * 16 + fq * 4;
    const float4 b4 = *(const float4*)(bias + n0 + n_loc);
    const int plane = n_loc >> 6, d = n_loc & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wm * 64 + j * 16 + fr;
      uint2 p;
      p.x = pack_op2(acc[i][j][0] + b4.x, acc[i][j][1] + b4.y);
      p.y = pack_op2(acc[i][j][2] + b4.z, acc[i][j][3] + b4.w);
      *(uint2*)(smem + plane * 32768 + row * 128 + (((d >> 3) ^ (row & 7)) << 4) + (d & 4) * 2) = p;
    }
  }
  __syncthreads();

  // ---- attention of (sequence s of the tile, query block qb) on wave w = s * KB + qb: attention_kernel's arithmetic ----
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef short v4s __attribute__((ext_vector_type(4)));
  const char* Qs = smem;
  const char* Ks = smem + 32768;
  const char* Vs = smem + 65536;
  const int s = wave / KB, qb = wave % KB;
  if (s * KB * 16 >= 256) return;                         // KB = 16: one sequence, all waves busy; never taken otherwise
  const int r0 = s * (KB * 16);                           // first tile row of the sequence
  bf16x8 qf[2];
  {
    const int qrow = r0 + qb * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const bf16x8*)(Qs + qrow * 128 + (((kk * 4 + fq) ^ (qrow & 7)) << 4));
  }
  f32x4 st[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int krow = r0 + kb * 16 + fr;
    st[kb] = mfma_op16(*(const bf16x8*)(Ks + krow * 128 + ((fq ^ (krow & 7)) << 4)), qf[0], (f32x4){0.f, 0.f, 0.f, 0.f});
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int krow = r0 + kb * 16 + fr;
    st[kb] = mfma_op16(*(const bf16x8*)(Ks + krow * 128 + (((4 + fq) ^ (krow & 7)) << 4)), qf[1], st[kb]);
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][r]);
  mx = rows4_max(mx);
  const f32x2 l2e = {1.44269504088896341f, 1.44269504088896341f};
  const float mneg1 = -mx * 1.44269504088896341f;
  const f32x2 mneg = {mneg1, mneg1};
  f32x2 sum2 = {0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const f32x2 a = __builtin_elementwise_fma((f32x2){st[kb][0], st[kb][1]}, l2e, mneg);
    const f32x2 b = __builtin_elementwise_fma((f32x2){st[kb][2], st[kb][3]}, l2e, mneg);
    const f32x2 ea = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const f32x2 eb = {__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
    st[kb] = (f32x4){ea[0], ea[1], eb[0], eb[1]};
    sum2 += ea;
    sum2 += eb;
  }
  float sum = sum2[0] + sum2[1];
  sum = rows4_sum(sum);
  const float inv = 1.0f / sum;

  f32x4 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < KB / 2; ++c) {
    union { bf16x8 v; uint32_t u[4]; } pf;
    const f32x4 lo = st[2 * c], hi = st[2 * c + 1];
    pf.u[0] = pack_op2(lo[0], lo[1]);
    pf.u[1] = pack_op2(lo[2], lo[3]);
    pf.u[2] = pack_op2(hi[0], hi[1]);
    pf.u[3] = pack_op2(hi[2], hi[3]);
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      union { bf16x8 v; uint2 h2[2]; } vf;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int krow = r0 + (2 * c + hh) * 16 + fq * 4 + (fr >> 2);
        const int dcol = db * 16 + (fr & 3) * 4;
        const char* a = Vs + krow * 128 + (((dcol >> 3) ^ (krow & 7)) << 4) + ((dcol >> 2) & 1) * 8;
        vf.h2[hh] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(
                                                  (__attribute__((address_space(3))) char*)a)));
      }
      o[db] = mfma_op16(vf.v, pf.v, o[db]);
    }
  }
  // the query's token: operand row m' = (b*C + c)*R + r  ->  context row (b*R + r)*C + c (ordinary token order)
  const int64_t mp = (int64_t)m0 + r0 + qb * 16 + fr;
  if (mp < m_real) {
    const int64_t seq = mp / R;
    const int rr = (int)(mp - seq * R);
    const int64_t b = seq / C;
    const int cc = (int)(seq - b * C);
    bf16_t* dst = ctx + ((b * R + rr) * C + cc) * ld_ctx + tile_n * 64 + fq * 4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      uint2 p;
      p.x = pack_op2(o[db][0] * inv, o[db][1] * inv);
      p.y = pack_op2(o[db][2] * inv, o[db][3] * inv);
      *(uint2*)(dst + db * 16) = p;
    }
  }
}

bool colattn_ok(int R, int d_model, int n_heads) {
  static const int on = [] { const char* e = getenv("PGIBBS_MSA_COLFUSE"); return e ? atoi(e) : 1; }();
  return on && (R == 32 || R == 64 || R == 128 || R == 256) && d_model % 64 == 0 && n_heads * 64 == d_model;
}

// X: LayerNorm output rows in column-major token order, [round_up(B*C*R, 256)][d] (rows beyond B*C*R: anything finite);
// W: [H*192][d] with head h's q, k, v rows at h*192 .. (launch_headmajor_qkv); ctx: [B*R*C][ld_ctx] in token order.
int launch_gemm_colattn(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, bf16_t* ctx, int B, int R, int C, int H,
                        int d, int ld_ctx) {
  if (!colattn_ok(R, d, H) || d % 64 || d < 64) return fail(1, "gemm_colattn: shape");
  const int64_t m_real = (int64_t)B * C * R;
  const int64_t Mp = (m_real + 255) / 256 * 256;
  if (Mp / 256 * H > 0x7fffffff) return fail(1, "gemm_colattn: too many tiles");
  const int tiles_n = H, n_tiles = (int)(Mp / 256) * H;
  dim3 grid(n_tiles), block(1024);
#define PG_CA(KBV)                                                                                                            \
  hipLaunchKernelGGL((gemm_colattn_kernel<KBV, 4>), grid, block, 0, s, X, W, bias, ctx, d, d, d, ld_ctx, tiles_n, n_tiles, R, C, m_real)
  if (R == 32) PG_CA(2);
  else if (R == 64) PG_CA(4);
  else if (R == 128) PG_CA(8);
  else PG_CA(16);
#undef PG_CA
  PG_HIP(hipGetLastError());
  return 0;
}

// [q | k | v] projection rows ([3*H*64][K], q rows first) -> per head [q_h | k_h | v_h] ([H*192][K]); bias likewise
__global__ __launch_bounds__(256) void headmajor_rows_kernel(const bf16_t* __restrict__ w, const float* __restrict__ b, bf16_t* __restrict__ w2,
                                                            float* __restrict__ b2, int H, int K) {
  const int row2 = blockIdx.x;                       // destination row: h*192 + part*64 + i
  const int h = row2 / 192, part = (row2 % 192) / 64, i = row2 % 64;
  const int row = part * H * 64 + h * 64 + i;
  for (int c = threadIdx.x; c < K; c += 256) w2[(size_t)row2 * K + c] = w[(size_t)row * K + c];
  if (threadIdx.x == 0) b2[row2] = b[row];
}
int launch_headmajor_qkv(hipStream_t s, const bf16_t* w, const float* b, bf16_t* w2, float* b2, int H, int K) {
  hipLaunchKernelGGL(headmajor_rows_kernel, dim3(H * 192), dim3(256), 0, s, w, b, w2, b2, H, K);
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
