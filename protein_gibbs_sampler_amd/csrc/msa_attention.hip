// Tied row attention of the MSA Transformer (fair-esm RowSelfAttention), reached by the reference
// through `self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_msa_sampler.py:136,236):
//   A[h,i,j] = scale * sum_r sum_d q[r,i,h,d] k[r,j,h,d]      (scale = 64^-0.5 / sqrt(R), one map per head for ALL rows)
//   P = softmax_j(A);   ctx[r,i,h,:] = sum_j P[h,i,j] v[r,j,h,:]
// (Column attention is the plain fused attention kernel over strided sequences: attention.hip.)
//
// One workgroup = (msa b, head h, NW*16 queries); wave w owns 16 of them.  Pass 1 streams K_r (C x 64) through LDS for
// r = 0..R-1 and accumulates the S^T blocks of the wave's 16 queries in registers with v_mfma_f32_16x16x32_bf16
// (the same lane-local layout as attention.hip: a lane holds one query's scores for 4 keys of every 16-key block),
// then one exact softmax; pass 2 streams V_r through LDS the same way (V^T fragments via ds_read_b64_tr_b16) and emits
// ctx for every row with the same P fragments.
// Every workgroup of a (msa, head) re-streams all R tiles, so the workgroup is made as wide as the registers allow:
// NW = 9 waves cover C = 257 columns (bos + 256) in 2 workgroups instead of 5 (the 5th held a single query).
// The tiles alternate between two LDS buffers (one barrier per row) and are fetched into registers two rows ahead
// (one for the widest tiles); measured at 64 MSAs x 32 x 257: 1.79 -> 1.31 ms per layer, of which the strided K / V / Q
// tile loads (128 contiguous bytes per token row) are 0.55 ms and the MFMAs 0.1 ms;
// the query chunks of one (msa, head) are placed on one XCD so the re-streamed tiles hit its L2.
#include "kernels.h"

PG_OPS_BEGIN

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MAXKB, int NW, int mode>
__global__ __launch_bounds__(NW * 64) void msa_row_attention_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int R, int C, int H, int ld_qkv, int ld_ctx, int k_off,
    int v_off, float scale, int n_qblk, int n_bh, int n_rc, float* __restrict__ partial) {
  // mode 0: fused (both passes over all R rows).  Split-R (few (msa, head) pairs, e.g. generate_single with B = 1):
  // mode 1: pass 1 over the row chunk rc only, partial S^T -> partial[bh][rc][query][key];
  // mode 3: (one workgroup per (msa, head, query block)) S = sum of the n_rc partial maps in fixed order, softmax,
  //         P as bf16 MFMA fragments -> pfrag (16 B per lane and fragment, stored behind the partial maps);
  // mode 2: pass 2 over the row chunk rc only with the P fragments of mode 3.
  constexpr int tpad = MAXKB * 16;
  constexpr int NT = NW * 64;                      // threads
  constexpr int NIT = (tpad * 8 + NT - 1) / NT;    // tile items per thread: one uint4 = 8 d of one key
  constexpr int AHEAD = (MAXKB <= 18 && NIT <= 9) ? 2 : 1;   // rows fetched ahead into registers (register budget)
  constexpr int BUF = tpad * 128;                  // one tile buffer: tpad key rows of 128 B (K_r in pass 1, V_r in pass 2)
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  // mode 3 is launched with one wave per workgroup (the reduction has no LDS phase; this spreads it over the chip):
  // blockIdx carries the wave slot
  const int tid = threadIdx.x, lane = tid & 63, wave = mode == 3 ? (int)(blockIdx.x % NW) : tid >> 6;
  const int blk = mode == 3 ? (int)(blockIdx.x / NW) : (int)blockIdx.x;
  // XCD-aware mapping (block b runs on XCD b % 8): the query chunks of one (msa, head) get consecutive slots of ONE
  // XCD, so the K_r / V_r tiles they all stream are served by that XCD's L2 after the first touch.
  int qblk, bh;
  const int per_rc = n_bh * n_qblk;
  const int rc = blk / per_rc;                       // row chunk (0 in fused mode)
  {
    const int bidx = blk - rc * per_rc;
    const int xcd = bidx & 7, slot = bidx >> 3;
    const int full = (n_bh / 8) * 8;                 // (msa, head) pairs covered by the XCD-aligned part of the grid
    if (bidx < full * n_qblk) {
      bh = (slot / n_qblk) * 8 + xcd;
      qblk = slot % n_qblk;
    } else {                                         // remainder: plain order
      const int rest = bidx - full * n_qblk;
      bh = full + rest / n_qblk;
      qblk = rest % n_qblk;
    }
  }
  const int rsz = (R + n_rc - 1) / n_rc;
  const int r_lo = mode == 0 ? 0 : rc * rsz;
  const int r_hi = mode == 0 ? R : ((rc + 1) * rsz < R ? (rc + 1) * rsz : R);
  const int b = bh / H, h = bh % H;
  const bf16_t* base = qkv + (size_t)b * R * C * ld_qkv + h * 64;   // row (r*C + i)
  const int fr = lane & 15, fq = lane >> 4;
  const int q0 = qblk * (NW * 16) + wave * 16;
  const bool active = q0 < C;                                      // wave-uniform
  int qrow = q0 + fr;
  if (qrow >= C) qrow = C - 1;

  f32x4 st[MAXKB];
#pragma unroll
  for (int kb = 0; kb < MAXKB; ++kb) st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- pass 1: scores summed over rows.  Tile r goes to LDS buffer r & 1; the registers it came from are refilled with
  // row r + 2 right away, so a tile has two rows of MFMAs to arrive.
  struct KRegs { uint4 k[NIT]; };                    // one row's K tile share
  KRegs kA, kB;
  auto load_k = [&](KRegs& g, int r, int off) {
    const bf16_t* rb = base + (size_t)r * C * ld_qkv;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT, row = i >> 3, c = i & 7;
      g.k[it] = make_uint4(0, 0, 0, 0);
      if (i < tpad * 8 && row < C) g.k[it] = *(const uint4*)(rb + (size_t)row * ld_qkv + off + c * 8);
    }
  };
  auto score_row = [&](KRegs& g, int r) {
    char* Ks = smem + (r & 1) * BUF;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT, row = i >> 3, c = i & 7;
      if (i < tpad * 8) *(uint4*)(Ks + row * 128 + ((c ^ (row & 7)) << 4)) = g.k[it];
    }
    const bf16_t* qp = base + ((size_t)r * C + qrow) * ld_qkv + fq * 8;     // in flight across the barrier
    const bf16x8 qf0 = *(const bf16x8*)qp, qf1 = *(const bf16x8*)(qp + 32);
    __syncthreads();                                  // tile r visible; everybody is done with tile r - 1 (other buffer)
    if (r + AHEAD < r_hi) load_k(g, r + AHEAD, k_off);
    if (active) {
      // the two k-halves of a key block accumulate into the same registers: keep them MAXKB MFMAs apart
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb) {
          const int krow = kb * 16 + fr;
          const bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + (((kk * 4 + fq) ^ (krow & 7)) << 4));
          st[kb] = mfma_op16(kf, kk ? qf1 : qf0, st[kb]);
        }
      }
    }
  };
  if (mode < 2) {
    if (r_lo < r_hi) load_k(kA, r_lo, k_off);
    if (AHEAD == 2) {
      if (r_lo + 1 < r_hi) load_k(kB, r_lo + 1, k_off);
      for (int r = r_lo; r < r_hi; r += 2) {
        score_row(kA, r);
        if (r + 1 < r_hi) score_row(kB, r + 1);
      }
    } else {
      for (int r = r_lo; r < r_hi; ++r) score_row(kA, r);
    }
  }

  if (mode == 1) {          // partial score map of this row chunk: lane stores its 4 keys x MAXKB blocks for query q0+fr
    if (active && q0 + fr < C) {
      float* dst = partial + (((size_t)bh * n_rc + rc) * C + (q0 + fr)) * tpad + fq * 4;
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) *(float4*)(dst + kb * 16) = make_float4(st[kb][0], st[kb][1], st[kb][2], st[kb][3]);
    }
    return;
  }
  if (mode == 3 && active) {
    const int qq = (q0 + fr < C) ? q0 + fr : C - 1;
    for (int c2 = 0; c2 < n_rc; ++c2) {
      const float* src = partial + (((size_t)bh * n_rc + c2) * C + qq) * tpad + fq * 4;
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) {
        const float4 v = *(const float4*)(src + kb * 16);
        st[kb][0] += v.x; st[kb][1] += v.y; st[kb][2] += v.z; st[kb][3] += v.w;
      }
    }
  }
  union PF { bf16x8 v; uint32_t u[4]; uint4 q; };
  PF pf[MAXKB / 2];
  // fragment store of this wave: [(bh, qblk, wave)][c][lane]
  uint4* pfrag = (uint4*)(partial + (size_t)n_bh * n_rc * C * tpad) + (((size_t)bh * n_qblk + qblk) * NW + wave) * (MAXKB / 2) * 64 + lane;
  if (mode == 2) {
#pragma unroll
    for (int c = 0; c < MAXKB / 2; ++c) pf[c].q = pfrag[c * 64];
  } else {
    // ---- softmax over keys ---------------------------------------------------------------------
    constexpr float LOG2E = 1.44269504088896341f;
    float mx = -3.0e38f;
    int tl = C - fq * 4;
  #pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
  #pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        st[kb][r4] *= scale;
        if (kb >= MAXKB - 6 && kb * 16 + r4 >= tl) st[kb][r4] = -3.0e38f;
        mx = fmaxf(mx, st[kb][r4]);
      }
    }
    mx = rows4_max(mx);
    const float mneg = -mx * LOG2E;
    float sum = 0.f;
  #pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
  #pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float e = __builtin_amdgcn_exp2f(fmaf(st[kb][r4], LOG2E, mneg));
        st[kb][r4] = e;
        sum += e;
      }
    }
    sum = rows4_sum(sum);
    const float inv = 1.0f / sum;
  #pragma unroll
    for (int c = 0; c < MAXKB / 2; ++c) {
      const f32x4 lo = st[2 * c], hi = st[2 * c + 1];
      pf[c].u[0] = pack_op2(lo[0] * inv, lo[1] * inv);
      pf[c].u[1] = pack_op2(lo[2] * inv, lo[3] * inv);
      pf[c].u[2] = pack_op2(hi[0] * inv, hi[1] * inv);
      pf[c].u[3] = pack_op2(hi[2] * inv, hi[3] * inv);
    }
    if (mode == 3) {
#pragma unroll
      for (int c = 0; c < MAXKB / 2; ++c) pfrag[c * 64] = pf[c].q;
      return;
    }
  }

  // ---- pass 2: ctx[r] = P . V_r for every row.  V_r tiles are staged exactly like the K_r tiles (row-major, swizzled,
  // two buffers, fetched AHEAD rows ahead); the V^T fragments of the PV MFMAs come out of them through the transposing
  // LDS read ds_read_b64_tr_b16 (see attention.hip) -- the former transposition pass was most of this kernel's VALU work.
  typedef short v4s __attribute__((ext_vector_type(4)));
  auto apply_row = [&](KRegs& g, int r, int slot) {
    char* Vs = smem + slot * BUF;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT, row = i >> 3, c = i & 7;
      if (i < tpad * 8) *(uint4*)(Vs + row * 128 + ((c ^ (row & 7)) << 4)) = g.k[it];
    }
    __syncthreads();
    if (r + AHEAD < r_hi) load_k(g, r + AHEAD, v_off);
    if (active) {
      f32x4 o[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < MAXKB / 2; ++c) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          union { bf16x8 v; uint2 h2[2]; } vf;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int krow = (2 * c + hh) * 16 + fq * 4 + (fr >> 2);
            const int dcol = db * 16 + (fr & 3) * 4;
            const char* a = Vs + krow * 128 + (((dcol >> 3) ^ (krow & 7)) << 4) + ((dcol >> 2) & 1) * 8;
            vf.h2[hh] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(
                                                      (__attribute__((address_space(3))) char*)a)));
          }
          o[db] = mfma_op16(vf.v, pf[c].v, o[db]);
        }
      }
      const int q = q0 + fr;
      if (q < C) {
        bf16_t* dst = ctx + ((size_t)(b * R + r) * C + q) * ld_ctx + h * 64 + fq * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          uint2 p;
          p.x = pack_op2(o[db][0], o[db][1]);
          p.y = pack_op2(o[db][2], o[db][3]);
          *(uint2*)(dst + db * 16) = p;
        }
      }
    }
  };
  __syncthreads();                                    // pass 1's last tile (either buffer) fully consumed
  if (r_lo < r_hi) load_k(kA, r_lo, v_off);
  if (AHEAD == 2) {
    if (r_lo + 1 < r_hi) load_k(kB, r_lo + 1, v_off);
    for (int r = r_lo; r < r_hi; r += 2) {
      apply_row(kA, r, 0);
      if (r + 1 < r_hi) apply_row(kB, r + 1, 1);
    }
  } else {
    for (int r = r_lo; r < r_hi; ++r) apply_row(kA, r, (r - r_lo) & 1);
  }
}

// split-R geometry, shared by the launcher and by the engine's scratch sizing
static void row_split_geometry(int n_bh, int obh, int R, int C, int& nw, int& n_qblk, int& n_rc, int& kb) {
  const int chunks = (C + 15) / 16;
  nw = C <= 64 ? 4 : (C <= 288 ? 9 : 8);
  n_qblk = (chunks + nw - 1) / nw;
  n_rc = 1;
  if (obh * ((C + 63) / 64) < 384 && obh * n_qblk < 384 && R >= 8) {      // few (msa, head, 64-query) units: the chip would idle
    n_rc = 512 / (obh * n_qblk);                      // two whole rounds of one workgroup per CU (measured best: 512 vs 768)
    if (n_rc > R / 4) n_rc = R / 4;
    if (n_rc < 1) n_rc = 1;
  }
  // a rung for every even block count (round 4; PGIBBS_ATTN_LADDER=0: the coarse ladder 2 4 8 12 18 24 30 36): the key blocks beyond C
  // are zero-filled and masked -- exact zeros in every sum, so the rung does not change the bits, only the wasted work
  static const int fine = [] { const char* e = getenv("PGIBBS_ATTN_LADDER"); return e ? atoi(e) : 1; }();
  static const int kbs_fine[] = {2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36};
  static const int kbs_coarse[] = {2, 4, 8, 12, 18, 24, 30, 36};
  const int* kbs = fine ? kbs_fine : kbs_coarse;
  const int n_kbs = fine ? 18 : 8;
  kb = 36;
  for (int i = 0; i < n_kbs; ++i)
    if (C <= kbs[i] * 16) { kb = kbs[i]; break; }
  // One exception, measured (profiles/r04_attention_key_block_ladder_ab.txt): one template of 32 x 301 (288 workgroups, split-R form)
  // takes 0.97 ms on the 20-block rung against 0.88 on the 24-block one, while bigger grids gain 10 %.  (The 20-block rung's 80 KB of
  // LDS lets two workgroups share a CU; whether that is the cause was not established.)  Grids below two rounds stay on 24 blocks.
  if (kb == 20 && (long)n_bh * n_qblk * n_rc < 512) kb = 24;
  (void)n_bh;
}
// bytes of fp32 scratch the split-R form needs for B alignments (0: the shape does not split)
size_t msa_row_split_scratch_bytes(int B, int R, int C, int H, int order_bh) {
  if (C > 576) return 0;
  int nw, n_qblk, n_rc, kb;
  row_split_geometry(B * H, order_bh > 0 ? order_bh : B * H, R, C, nw, n_qblk, n_rc, kb);
  if (n_rc <= 1) return 0;
  return (size_t)B * H * n_rc * C * (kb * 16) * 4 + (size_t)B * H * n_qblk * nw * (kb / 2) * 1024;
}

int launch_msa_row_attention_bf16(hipStream_t s, const bf16_t* qkv, bf16_t* ctx, int B, int R, int C, int H, int ld_qkv,
                                  int ld_ctx, int k_off, int v_off, float scale, float* partial, size_t partial_bytes,
                                  int order_bh) {
  if (B == 0 || R == 0) return 0;
  if (C <= 0) return fail(1, "row attention: empty alignment");
  if (C > 576) return fail(5, "row attention: alignments wider than 576 token columns (<cls> + 575 residues) take the fp32-scores path");
  const int n_bh = B * H;
  // workgroup width: 4 waves up to 64 columns, beyond that the widest the register budget of the score fragments allows
  // (9 waves up to 288 keys, 8 beyond).  The row loop is split over workgroups when the (msa, head, query-chunk) grid alone
  // cannot fill the chip; the number of row chunks fixes the order of the sum over alignment rows, so it is derived from
  // order_bh -- the (msa, head) count of the JOB, or of ONE template in a batched generate_single -- not from this call's share
  int nw, n_qblk, n_rc, kb_;
  row_split_geometry(n_bh, order_bh > 0 ? order_bh : n_bh, R, C, nw, n_qblk, n_rc, kb_);
  if (!partial) n_rc = 1;
#define PG_ROWATT_K(KB, NWV, MODE, GRID, BLOCK)                                                                          \
  hipLaunchKernelGGL((msa_row_attention_kernel<KB, NWV, MODE>), GRID, BLOCK, 0, s, qkv, ctx, R, C, H, ld_qkv, ld_ctx,     \
                     k_off, v_off, scale, n_qblk, n_bh, n_rc, partial)
#define PG_ROWATT(KB, NWV)                                                                                              \
  else if (kb_ == KB) {                                                                                                 \
    if (n_rc > 1 && (size_t)n_bh * n_rc * C * (KB * 16) * 4 + (size_t)n_bh * n_qblk * nw * (KB / 2) * 1024 > partial_bytes) n_rc = 1; \
    const dim3 block(NWV * 64), grid((unsigned)(n_bh * n_qblk * n_rc));                                                 \
    if (n_rc == 1) {                                                                                                    \
      PG_ROWATT_K(KB, NWV, 0, grid, block);                                                                             \
    } else {                                                                                                            \
      PG_ROWATT_K(KB, NWV, 1, grid, block);                                                                             \
      PG_ROWATT_K(KB, NWV, 3, dim3((unsigned)(n_bh * n_qblk * NWV)), dim3(64));                                         \
      PG_ROWATT_K(KB, NWV, 2, grid, block);                                                                             \
    }                                                                                                                   \
  }
  if (false) {}
  PG_ROWATT(2, 4) PG_ROWATT(4, 4) PG_ROWATT(6, 9) PG_ROWATT(8, 9) PG_ROWATT(10, 9) PG_ROWATT(12, 9) PG_ROWATT(14, 9) PG_ROWATT(16, 9)
  PG_ROWATT(18, 9) PG_ROWATT(20, 8) PG_ROWATT(22, 8) PG_ROWATT(24, 8) PG_ROWATT(26, 8) PG_ROWATT(28, 8) PG_ROWATT(30, 8) PG_ROWATT(32, 8)
  PG_ROWATT(34, 8) PG_ROWATT(36, 8)
#undef PG_ROWATT
#undef PG_ROWATT_K
  else {
    return fail(5, "row attention: alignments wider than 576 token columns (<cls> + 575 residues) take the fp32-scores path");
  }
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
