// Fused multi-head self-attention for one sequence per workgroup (ESM-1b layers):
//   ctx[b, t, h*64:(h+1)*64] = softmax_j( q[b,t,h] . k[b,j,h] ) @ v[b,:,h]        (q pre-scaled by 64^-0.5)
// This is the attention inside fair-esm's ProteinBertModel that the reference reaches through
// `self.model.model(batch)["logits"]` (/root/reference/src/pgen/esm_sampler.py:223).
//
// CDNA4 mapping (head dim is 64 in ESM-1b and MSA-1b):
//   * grid = B*H workgroups of 4 waves; K and V (both row-major, XOR-swizzled 16-B chunks) live in LDS for
//     the whole sequence (T <= 576: 2 x 72 KB) and are shared by all query blocks; longer sequences use
//     attention_long_kernel (288-key tiles, online softmax).
//   * per wave, 16 queries at a time: S^T = K.Q^T with v_mfma_f32_16x16x32_bf16 ("swapped" product), so a
//     lane holds, for ONE query (lane & 15), 4 consecutive keys of every 16-key block: the whole score
//     row is lane-local except for a 4-lane (xor 16, 32) shuffle reduction -> exact (non-online) softmax
//     in registers, no LDS round trip for P.
//   * the K-slot order of the PV contraction is free, so it is chosen to be exactly the order the lane
//     already holds P in (two 16-key blocks per 32-wide MFMA step); the matching V^T fragment (4 + 4 keys of one d per
//     lane) comes straight out of the row-major V tile through gfx950's transposing LDS read ds_read_b64_tr_b16 -- no
//     transposition pass when the tile is staged (that pass and its 4-byte LDS writes were 16 % of the kernel).
//   * O^T = V^T.P^T, so a lane ends with 4 consecutive d for one query -> 8-byte row-major stores.
#include <stdlib.h>

#include "kernels.h"

PG_OPS_BEGIN

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// MAXKB = number of 16-key blocks computed (T <= 16*MAXKB), even.  The dispatch ladder guarantees
// T > 16*(MAXKB-6), so only the last 6 blocks can hold masked (>= T) keys.
typedef short v4s __attribute__((ext_vector_type(4)));

// Phase timestamps for tools/probes/attention_phases.hip (compiled out of the library): lane 0 of every wave records the
// shader clock at the phase boundaries of its first query blocks.
#ifdef PG_ATT_PROF
__device__ unsigned long long* pg_att_prof;      // [workgroup][wave][8 slots][8 stamps]
#define PG_T(slot, i)                                                                                              \
  do {                                                                                                             \
    if ((slot) < 8 && (threadIdx.x & 63) == 0)                                                                     \
      pg_att_prof[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot)) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PG_T(slot, i)
#endif

// BIASKV: ESM-1's extra bias_k / bias_v key (a template parameter: the extra staging branch and the runtime key count cost the
// config-2 kernel 8 % when they were runtime conditions)
// SPLIT (round 6): the grid's LAST workgroups each take 1 / split of one (sequence, head) pair's query blocks instead of a whole pair
// -- workgroups [0, split_from) whole pairs, workgroup split_from + u the blocks part, part + split, ... (part = u % split) of pair
// split_from + u / split -- so that the partial last round of a launch (a 32-chain shard: 640 pairs on 512 resident workgroups =
// one full round + a quarter-full one that takes as long) becomes short workgroups that fill the chip.  A query block's arithmetic
// does not depend on which workgroup runs it: same bits.  The plain launch (SPLIT = false) is the kernel of rounds 1-5, unchanged.
template <int MAXKB, bool PADMASK, bool BIASKV = false, bool SPLIT = false>
__global__ __launch_bounds__(256, (MAXKB <= 18 ? 2 : 1)) void attention_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int T,
                                                       int H, int ld_qkv_, int ld_ctx_, int k_off, int v_off,
                                                       SeqLayout sl, const int32_t* __restrict__ key_tok, int pad_idx,
                                                       const bf16_t* __restrict__ bias_kv, int split_from = 0, int split = 1) {
  __shared__ __attribute__((aligned(16))) char smem[2 * MAXKB * 16 * 128 + (PADMASK ? MAXKB * 16 : 0)];
  char* Ks = smem;
  char* Vs = smem + MAXKB * 16 * 128;          // V rows, same layout as K: row*128 + ((chunk ^ (row & 7)) << 4)
  char* padf = smem + 2 * MAXKB * 16 * 128;    // PADMASK: one byte per key, 1 = this key's token is <pad>

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  PG_T(7, 0);
  // sequence `seq` = token rows row0 + t*row_step (ESM: contiguous rows of chain b; MSA column attention: the R rows
  // of one column, C token-rows apart)
  int pair = blockIdx.x, qb_first = wave, qb_step = 4;
  if (SPLIT && (int)blockIdx.x >= split_from) {
    const int u = blockIdx.x - split_from;
    pair = split_from + u / split;
    qb_first = u % split + split * wave;                     // blocks part, part + split, ... dealt to the waves in turn
    qb_step = 4 * split;
  }
  const int seq = pair / H, h = pair % H;
  const size_t row0 = (size_t)(seq / sl.inner_count) * sl.outer_rows + (size_t)(seq % sl.inner_count) * sl.inner_rows;
  const size_t ld_qkv = (size_t)ld_qkv_ * sl.row_step, ld_ctx = (size_t)ld_ctx_ * sl.row_step;
  const bf16_t* base = qkv + row0 * ld_qkv_ + h * 64;
  // ESM-1 (add_bias_kv): key T is the learned bias_k / bias_v of this head -- one more key, attended by every query, never masked
  const int Tk = T + (BIASKV ? 1 : 0);
  // All MAXKB key blocks are computed unconditionally: K rows / V^T columns past T are zero-filled and
  // their scores are masked, so no wave-uniform branches (and no dynamic register indexing) are needed.
  constexpr int nkc = MAXKB / 2;
  constexpr int tpad = MAXKB * 16;

  // ---- stage K and V (swizzled rows).  All global loads are issued before the first LDS write so a block pays
  //      ~one memory round trip, not one per loop iteration.
  {
    constexpr int NIT = (MAXKB * 16 * 8 + 255) / 256;       // one uint4 (8 d of one key) per item
    uint4 kreg[NIT], vreg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int row = i >> 3, c = i & 7;
      kreg[it] = make_uint4(0, 0, 0, 0);
      vreg[it] = make_uint4(0, 0, 0, 0);
      if (i < tpad * 8 && row < T) {
        kreg[it] = *(const uint4*)(base + (size_t)row * ld_qkv + k_off + c * 8);
        vreg[it] = *(const uint4*)(base + (size_t)row * ld_qkv + v_off + c * 8);
      } else if (BIASKV && row == T) {
        kreg[it] = *(const uint4*)(bias_kv + h * 64 + c * 8);
        vreg[it] = *(const uint4*)(bias_kv + (H + h) * 64 + c * 8);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      if (i < tpad * 8) {
        const int row = i >> 3, c = i & 7;
        *(uint4*)(Ks + row * 128 + ((c ^ (row & 7)) << 4)) = kreg[it];
        *(uint4*)(Vs + row * 128 + ((c ^ (row & 7)) << 4)) = vreg[it];
      }
    }
  }
  if (PADMASK) {
    // the <pad> flags of the sequence's keys, once per workgroup (read per key inside the score loop they were 72 dependent global
    // loads per lane and query block: 292 bytes of scratch, a ragged batch's attention 5x the time of a full one)
    for (int key = tid; key < MAXKB * 16; key += 256)
      padf[key] = (key < T && key_tok[row0 + (size_t)key * sl.row_step] == pad_idx) ? 1 : 0;
  }
  PG_T(7, 1);
  __syncthreads();
  PG_T(7, 2);

  const int fr = lane & 15, fq = lane >> 4;
  const int nqb = (T + 15) >> 4;  // query blocks of 16
  // Q fragment (MFMA B operand): query fr, d = kk*32 + fq*8 .. +7; the next block's fragment is prefetched while
  // the current one is computed (a wave has nothing else to cover a global round trip with)
  bf16x8 qf[2], qn[2];
  auto load_q = [&](int qb, bf16x8 (&dst)[2]) {
    int qrow = qb * 16 + fr;
    if (qrow >= T) qrow = T - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) dst[kk] = *(const bf16x8*)(base + (size_t)qrow * ld_qkv + kk * 32 + fq * 8);
  };
  if (qb_first < nqb) load_q(qb_first, qf);
  for (int qb = qb_first; qb < nqb; qb += qb_step) {
    if (qb + qb_step < nqb) load_q(qb + qb_step, qn);
    PG_T(qb >> 2, 0);

    // S^T blocks: st[kb][r] = S[query fr][key kb*16 + fq*4 + r]
    f32x4 st[MAXKB];
    {
      // K fragments are fetched one chunk (CH key blocks) ahead of the MFMAs that use them: with 2 waves per SIMD
      // the LDS latency must be covered inside the wave (PMC: 44 % of wave cycles were s_waitcnt before this)
      constexpr int CH = (MAXKB % 6 == 0) ? 6 : (MAXKB % 4 == 0 ? 4 : 2);
      bf16x8 kbuf[2][CH][2];
      auto load_chunk = [&](int ch, bf16x8 (&dst)[CH][2]) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#if defined(PG_ATT_PROF) && PG_ATT_ABL == 1      /* ablation: one K fragment read per chunk instead of CH */
          const int krow = (ch * CH + (u > 0 ? 0 : u)) * 16 + fr;
          if (u > 0) { dst[u][0] = dst[0][0]; dst[u][1] = dst[0][1]; continue; }
#else
          const int krow = (ch * CH + u) * 16 + fr;
#endif
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) dst[u][kk] = *(const bf16x8*)(Ks + krow * 128 + (((kk * 4 + fq) ^ (krow & 7)) << 4));
        }
      };
      load_chunk(0, kbuf[0]);
#pragma unroll
      for (int ch = 0; ch < MAXKB / CH; ++ch) {
        if (ch + 1 < MAXKB / CH) load_chunk(ch + 1, kbuf[(ch + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        // the two d-halves of a key block as two sweeps over the chunk, so that no MFMA is issued right behind the one
        // producing its accumulator input (hipcc otherwise pairs them through one temporary register)
#pragma unroll
        for (int u = 0; u < CH; ++u)
          st[ch * CH + u] = mfma_op16(kbuf[ch & 1][u][0], qf[0], (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int u = 0; u < CH; ++u)
          st[ch * CH + u] = mfma_op16(kbuf[ch & 1][u][1], qf[1], st[ch * CH + u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    PG_T(qb >> 2, 1);
    // exact softmax over keys (lane-local + 4-lane reduction); exp(s - m) = exp2(s*log2e - m*log2e)
    float mx = -3.0e38f;
    int tl = Tk - fq * 4;                     // key kb*16 + fq*4 + r is padding iff kb*16 + r >= tl
    asm volatile("" : "+v"(tl));              // keep the compares inside the loop (no hoisted lane masks)
#pragma unroll
    for (int kb = MAXKB > 6 ? MAXKB - 6 : 0; kb < MAXKB; ++kb)
      if ((kb + 1) * 16 > Tk) {               // wave-uniform: only key blocks that reach past the last key are touched
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb * 16 + r >= tl) st[kb][r] = -3.0e38f;
      }
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][r]);      // -> v_max3_f32
    if (PADMASK) {
      // ragged batch (only reachable through the forward entry points; the Gibbs path never holds <pad>): keys that are
      // <pad> tokens are masked.  key_tok = the token buffer; the token of key t of this sequence sits where its qkv row does
      // (row0 + t * row_step).  Chains (ESM: contiguous rows) get -inf like fair-esm's key_padding_mask; the strided sequences
      // of the MSA Transformer's column attention get its finite fill of -10000 (an all-<pad> column then softmaxes to a uniform
      // row instead of NaN, exactly as fair-esm's ColumnSelfAttention does -- and NaN at a padded position would reach the real
      // ones through 0 * NaN in the next tied row attention).
      const float fill = sl.row_step == 1 ? -3.0e38f : -10000.0f;
      mx = -3.0e38f;
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) {
        const uint32_t f4 = *(const uint32_t*)(padf + kb * 16 + fq * 4);     // keys kb*16 + fq*4 .. +3
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if ((f4 >> (8 * r)) & 0xffu) st[kb][r] = fill;
          mx = fmaxf(mx, st[kb][r]);
        }
      }
    }
    mx = rows4_max(mx);
    // two scores per instruction (v_pk_fma_f32 / v_pk_add_f32): 284 instead of 418 VALU instructions per 16-query block,
    // 72 of them quarter-rate v_exp_f32
    const f32x2 l2e = {1.44269504088896341f, 1.44269504088896341f};
    const float mneg1 = -mx * 1.44269504088896341f;
    const f32x2 mneg = {mneg1, mneg1};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
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
    const float inv = 1.0f / sum;             // applied to O at the end: the lane's query (fr) is also its O column
    PG_T(qb >> 2, 2);

    // O^T[d][q] = sum_key V^T[d][key] * P^T[key][q]; K-slot (fq*8 + j) of chunk c <-> key (2c + (j>>2))*16 + fq*4 + (j&3)
    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      // V^T fragments fetched two 32-key chunks ahead of the PV MFMAs
      union VF { bf16x8 v; uint2 h[2]; };
      VF vbuf[3][4];
      auto load_v = [&](int c, VF (&dst)[4]) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#if defined(PG_ATT_PROF) && PG_ATT_ABL == 2      /* ablation: one V^T fragment read per chunk instead of 4 */
          if (db > 0) { dst[db] = dst[0]; continue; }
#endif
          // transposed LDS read (semantics probed on the device): the 16 lanes of a group point at 4 key rows x four 8-byte
          // pieces of 16 d (lane s -> row s>>2, piece s&3) and lane fr receives V[key0 .. key0+3][d = db*16 + fr]
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int krow = (2 * c + hh) * 16 + fq * 4 + (fr >> 2);
            const int dcol = db * 16 + (fr & 3) * 4;                                     // bf16 index inside the key row
            const char* a = Vs + krow * 128 + (((dcol >> 3) ^ (krow & 7)) << 4) + ((dcol >> 2) & 1) * 8;
            const v4s t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(
                (__attribute__((address_space(3))) char*)a));
            dst[db].h[hh] = __builtin_bit_cast(uint2, t);
          }
        }
      };
      load_v(0, vbuf[0]);
      if (nkc > 1) load_v(1, vbuf[1]);
#pragma unroll
      for (int c = 0; c < nkc; ++c) {
        if (c + 2 < nkc) load_v(c + 2, vbuf[(c + 2) % 3]);
        union { bf16x8 v; uint32_t u[4]; } pf;
        const f32x4 lo = st[2 * c], hi = st[2 * c + 1];
        pf.u[0] = pack_op2(lo[0], lo[1]);
        pf.u[1] = pack_op2(lo[2], lo[3]);
        pf.u[2] = pack_op2(hi[0], hi[1]);
        pf.u[3] = pack_op2(hi[2], hi[3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db] = mfma_op16(vbuf[c % 3][db].v, pf.v, o[db]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    PG_T(qb >> 2, 3);
    // store: lane holds O[q = qb*16 + fr][d = db*16 + fq*4 + r]
    const int q = qb * 16 + fr;
    if (q < T) {
      bf16_t* dst = ctx + row0 * ld_ctx_ + (size_t)q * ld_ctx + h * 64 + fq * 4;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        uint2 p;
        p.x = pack_op2(o[db][0] * inv, o[db][1] * inv);
        p.y = pack_op2(o[db][2] * inv, o[db][3] * inv);
        *(uint2*)(dst + db * 16) = p;
      }
    }
    qf[0] = qn[0];
    qf[1] = qn[1];
    PG_T(qb >> 2, 4);
  }
  PG_T(7, 3);
}

// ------------------------------------------------------------------------------------------------
// Long sequences (T > 576, up to the 1024-token limit of the learned position table): same MFMA formulation, keys
// processed in tiles of 288 with the online-softmax recurrence (running max m, running sum l, rescaled O).
// One workgroup = (sequence, head, 64 queries); wave w owns one 16-query block for the whole key loop, so the
// per-wave state is just O (16 regs) + m + l; every workgroup streams all K/V tiles of its head (L2-resident).
// ------------------------------------------------------------------------------------------------
// MAXKB = 16-key blocks per tile (even), OCC = workgroups per CU the register / LDS budget is set for: <18, 2> is the long-sequence
// kernel.  <10, 4> (160-key tiles, four workgroups per CU, 117 VGPRs) and <12, 3> were measured at config 2 in round 4 against
// attention_kernel: 9.1 / 11.1 ms per iteration against 7.0 (EXPERIMENTS.md) -- every 64-query workgroup re-stages the head's K / V.
template <int MAXKB, int OCC, bool PADMASK, bool BIASKV>
__global__ __launch_bounds__(256, OCC) void attention_long_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int T,
                                                               int H, int ld_qkv_, int ld_ctx_, int k_off, int v_off,
                                                               SeqLayout sl, int n_qchunk, const int32_t* __restrict__ key_tok_,
                                                               int pad_idx, const bf16_t* __restrict__ bias_kv_) {
  // the <pad> mask and ESM-1's bias key as template parameters (as in attention_kernel): as runtime conditions they kept the kernel
  // at 256 VGPRs with 110 dwords of scratch
  const int32_t* __restrict__ key_tok = PADMASK ? key_tok_ : nullptr;
  const bf16_t* __restrict__ bias_kv = BIASKV ? bias_kv_ : nullptr;
  constexpr int tpad = MAXKB * 16, nkc = MAXKB / 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * tpad * 128 + (PADMASK ? tpad : 0)];
  char* Ks = smem;
  char* Vs = smem + tpad * 128;
  char* padf = smem + 2 * tpad * 128;             // PADMASK: one byte per key of the tile, 1 = <pad> token (see attention_kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qc = blockIdx.x % n_qchunk, sh = blockIdx.x / n_qchunk;
  const int seq = sh / H, h = sh % H;
  const size_t row0 = (size_t)(seq / sl.inner_count) * sl.outer_rows + (size_t)(seq % sl.inner_count) * sl.inner_rows;
  const size_t ld_qkv = (size_t)ld_qkv_ * sl.row_step, ld_ctx = (size_t)ld_ctx_ * sl.row_step;
  const bf16_t* base = qkv + row0 * ld_qkv_ + h * 64;
  const int fr = lane & 15, fq = lane >> 4;
  const int q0 = qc * 64 + wave * 16;
  const bool active = q0 < T;                        // wave-uniform
  bf16x8 qf[2];
  {
    int qrow = q0 + fr;
    if (qrow >= T) qrow = T - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const bf16x8*)(base + (size_t)qrow * ld_qkv + kk * 32 + fq * 8);
  }
  f32x4 o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;
  constexpr float LOG2E = 1.44269504088896341f;
  const int Tk = T + (bias_kv ? 1 : 0);              // ESM-1: key T = this head's bias_k / bias_v (see attention_kernel)

  for (int k0 = 0; k0 < Tk; k0 += tpad) {
    __syncthreads();
    {   // stage this key tile: K and V rows, swizzled (as in attention_kernel)
      constexpr int NIT = (tpad * 8 + 255) / 256;
      uint4 kreg[NIT], vreg[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256, row = i >> 3, c = i & 7;
        kreg[it] = make_uint4(0, 0, 0, 0);
        vreg[it] = make_uint4(0, 0, 0, 0);
        if (i < tpad * 8 && k0 + row < T) {
          kreg[it] = *(const uint4*)(base + (size_t)(k0 + row) * ld_qkv + k_off + c * 8);
          vreg[it] = *(const uint4*)(base + (size_t)(k0 + row) * ld_qkv + v_off + c * 8);
        } else if (bias_kv && k0 + row == T) {
          kreg[it] = *(const uint4*)(bias_kv + h * 64 + c * 8);
          vreg[it] = *(const uint4*)(bias_kv + (H + h) * 64 + c * 8);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * 256, row = i >> 3, c = i & 7;
        if (i < tpad * 8) {
          *(uint4*)(Ks + row * 128 + ((c ^ (row & 7)) << 4)) = kreg[it];
          *(uint4*)(Vs + row * 128 + ((c ^ (row & 7)) << 4)) = vreg[it];
        }
      }
    }
    if (PADMASK) {
      for (int key = tid; key < tpad; key += 256)
        padf[key] = (k0 + key < T && key_tok[row0 + (size_t)(k0 + key) * sl.row_step] == pad_idx) ? 1 : 0;
    }
    __syncthreads();
    if (!active) continue;
    f32x4 st[MAXKB];
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
      st[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int krow = kb * 16 + fr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(Ks + krow * 128 + (((kk * 4 + fq) ^ (krow & 7)) << 4));
        st[kb] = mfma_op16(kf, qf[kk], st[kb]);
      }
    }
    float tmax = -3.0e38f;
    const int tl = Tk - k0 - fq * 4;             // key k0 + kb*16 + fq*4 + r is padding iff kb*16 + r >= tl
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kb * 16 + r >= tl) st[kb][r] = -3.0e38f;
        tmax = fmaxf(tmax, st[kb][r]);
      }
    if (PADMASK) {                               // <pad> keys of a ragged batch (see attention_kernel)
      const float fill = sl.row_step == 1 ? -3.0e38f : -10000.0f;
      tmax = -3.0e38f;
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) {
        const uint32_t f4 = *(const uint32_t*)(padf + kb * 16 + fq * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if ((f4 >> (8 * r)) & 0xffu) st[kb][r] = fill;
          tmax = fmaxf(tmax, st[kb][r]);
        }
      }
    }
    tmax = rows4_max(tmax);
    const float mn = fmaxf(m, tmax);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E);
    const float mneg = -mn * LOG2E;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(st[kb][r], LOG2E, mneg));
        st[kb][r] = e;
        psum += e;
      }
    psum = rows4_sum(psum);
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
    }
#pragma unroll
    for (int c = 0; c < nkc; ++c) {
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
        for (int hh = 0; hh < 2; ++hh) {                       // transposed LDS read, see attention_kernel
          const int krow = (2 * c + hh) * 16 + fq * 4 + (fr >> 2);
          const int dcol = db * 16 + (fr & 3) * 4;
          const char* a = Vs + krow * 128 + (((dcol >> 3) ^ (krow & 7)) << 4) + ((dcol >> 2) & 1) * 8;
          vf.h2[hh] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(
                                                    (__attribute__((address_space(3))) char*)a)));
        }
        o[db] = mfma_op16(vf.v, pf.v, o[db]);
      }
    }
  }
  const int q = q0 + fr;
  if (active && q < T) {
    const float inv = 1.0f / l;
    bf16_t* dst = ctx + row0 * ld_ctx_ + (size_t)q * ld_ctx + h * 64 + fq * 4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      uint2 p;
      p.x = pack_op2(o[db][0] * inv, o[db][1] * inv);
      p.y = pack_op2(o[db][2] * inv, o[db][3] * inv);
      *(uint2*)(dst + db * 16) = p;
    }
  }
}

int launch_attention_bf16(hipStream_t s, const bf16_t* qkv, bf16_t* ctx, int B, int T, int H, int ld_qkv, int ld_ctx,
                          int k_off, int v_off, const int32_t* key_tok, int pad_idx, const bf16_t* bias_kv) {
  SeqLayout sl = {1, T, 0, 1};
  return launch_attention_seq_bf16(s, qkv, ctx, B, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv);
}

int launch_attention_seq_bf16(hipStream_t s, const bf16_t* qkv, bf16_t* ctx, int64_t n_seq, int T, int H, int ld_qkv,
                              int ld_ctx, int k_off, int v_off, SeqLayout sl, const int32_t* key_tok, int pad_idx,
                              const bf16_t* bias_kv) {
  if (n_seq == 0) return 0;
  if (n_seq * H > 0x7fffffff) return fail(1, "attention: too many sequences");
  dim3 grid((unsigned)(n_seq * H)), block(256);
  // Round 6: split the pairs of a partial last round (see attention_kernel).  Whole-sequence kernels for chains (row_step 1) of at
  // least four query blocks, without <pad> mask / bias key (the Gibbs path).  Resident workgroups: two per CU up to 20 key blocks
  // (2 x 80 KB of LDS), one beyond.  PGIBBS_ATTN_SPLIT=0 switches it off.
  static const int split_on = [] { const char* e = getenv("PGIBBS_ATTN_SPLIT"); return e ? atoi(e) : 1; }();
  static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  int split_from = 0, split = 1;
  if (split_on && !key_tok && !bias_kv && sl.row_step == 1 && T >= 64 && T <= 576) {
    const int kb = (((T + 15) / 16) + 1) & ~1;            // the rung of the ladder below: key blocks, even
    const long pairs = n_seq * H, slots = (long)n_cu * (kb <= 20 ? 2 : 1);
    const long rem = pairs % slots;
    const int nqb = (T + 15) / 16;
    int sp = rem ? (int)(slots / rem) : 1;
    if (sp > 4) sp = 4;
    if (sp > nqb / 4) sp = nqb / 4;            // every part keeps at least one block per wave
    if (sp >= 2) {
      split = sp;
      split_from = (int)(pairs - rem);
      grid = dim3((unsigned)(split_from + rem * sp));
    }
  }
#define PG_ATT_SPLIT_LAUNCH(KB) hipLaunchKernelGGL((attention_kernel<KB, false, false, true>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv, split_from, split)
#define PG_ATT(KB)                                                                                             \
  else if (Tk <= KB * 16) {                                                                                    \
    if (bias_kv && key_tok) hipLaunchKernelGGL((attention_kernel<KB, true, true>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv); \
    else if (bias_kv) hipLaunchKernelGGL((attention_kernel<KB, false, true>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv); \
    else if (key_tok) hipLaunchKernelGGL((attention_kernel<KB, true>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv); \
    else if (split > 1) PG_ATT_SPLIT_LAUNCH(KB);                                                               \
    else hipLaunchKernelGGL((attention_kernel<KB, false>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv); \
  }
  const int Tk = T + (bias_kv ? 1 : 0);          // keys: the T tokens + ESM-1's bias_k / bias_v
  static const int fine_ladder = [] { const char* e = getenv("PGIBBS_ATTN_LADDER"); return e ? atoi(e) : 1; }();   // 0: the coarse ladder only
  if (T <= 0) return fail(1, "attention: empty sequence");
  // extra rungs for the forms without ESM-1's bias key (the Gibbs path, and ragged batches): a chain of 200 residues has 13 key blocks, not 18 --
  // the blocks beyond T are zero-filled and masked, i.e. pure waste (and exact zeros in every sum: the bits do not depend on the rung)
#define PG_ATT_PLAIN(KB)                                                                                       \
  else if (fine_ladder && !bias_kv && Tk <= KB * 16) {                                                         \
    if (key_tok) hipLaunchKernelGGL((attention_kernel<KB, true>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv); \
    else if (split > 1) PG_ATT_SPLIT_LAUNCH(KB);                                                               \
    else hipLaunchKernelGGL((attention_kernel<KB, false>), grid, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, key_tok, pad_idx, bias_kv); \
  }
  PG_ATT(2) PG_ATT(4) PG_ATT_PLAIN(6) PG_ATT(8) PG_ATT_PLAIN(10) PG_ATT(12) PG_ATT_PLAIN(14) PG_ATT_PLAIN(16) PG_ATT(18)
  PG_ATT_PLAIN(20) PG_ATT_PLAIN(22) PG_ATT(24) PG_ATT_PLAIN(26) PG_ATT_PLAIN(28) PG_ATT(30) PG_ATT_PLAIN(32) PG_ATT_PLAIN(34) PG_ATT(36)
#undef PG_ATT_PLAIN
#undef PG_ATT
#undef PG_ATT_SPLIT_LAUNCH
  else {
    const int n_qchunk = (T + 63) / 64;
    if (n_seq * H * n_qchunk > 0x7fffffff) return fail(1, "attention: too many sequences");
    const dim3 g((unsigned)(n_seq * H * n_qchunk));
#define PG_ATT_LONG(P, B) hipLaunchKernelGGL((attention_long_kernel<18, 2, P, B>), g, block, 0, s, qkv, ctx, T, H, ld_qkv, ld_ctx, k_off, v_off, sl, n_qchunk, key_tok, pad_idx, bias_kv)
    if (key_tok && bias_kv) PG_ATT_LONG(true, true);
    else if (key_tok) PG_ATT_LONG(true, false);
    else if (bias_kv) PG_ATT_LONG(false, true);
    else PG_ATT_LONG(false, false);
#undef PG_ATT_LONG
  }
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
