// PROBE, not part of libpgibbs.so (round 4).  attention_pipe_kernel<18>: block i+1's score MFMAs issued
// in the same scheduling region as block i's softmax arithmetic (sched_group_barrier interleave: 2 K fragment reads, 1 MFMA, 2
// v_exp_f32, ~5 VALU per step; verified in the ISA), the key-padding mask as the accumulators' initial value.  Bit-identical with
// attention_kernel<18> on 84.5 M values -- and NOT faster: 208.7 vs 207.8 us per launch at config 2 (MI355X, interleaved rounds).
// Timing ablations of the same kernel (-DPG_PIPE_ABL=n): one K fragment pair for all key blocks (no K LDS traffic) 179 us; the
// V^T fragments of chunk 0 for every chunk 201 us; no transcendentals 199 us.  So a wave's S / softmax serialisation is not what
// bounds the kernel (the two waves of a SIMD already overlap each other's phases); the largest single item is the K fragment
// traffic (every wave reads the whole K tile once per 16-query block).  Numbers: profiles/r04_attention_pipeline_probe.txt.
// Probe: attention_kernel<18> vs the software-pipelined attention_pipe_kernel<18> at BASELINE config 2 (256 chains x T = 258, 20
// heads): bitwise comparison of the outputs, then interleaved timing rounds (HIP events, 40 launches per round and kernel).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I protein_gibbs_sampler_amd/csrc tools/probes/attention_pipe_bench.hip -o build/att_pipe_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "attention.hip"

PG_OPS_BEGIN
// ------------------------------------------------------------------------------------------------
// Software-pipelined form of attention_kernel (round 4; no <pad> mask, no bias key: the Gibbs hot path).  The phase probe of
// round 2 put a wave's 16-query block at S = K.Q^T 2500 cycles (36 MFMAs + 36 LDS fragment reads), softmax 2800 (VALU: 72
// v_exp_f32 + ~210 other), P.V 2000, store 900 -- strictly one after the other, with the matrix pipe idle through the softmax
// and the VALU idle through the two products, and only ONE other wave on the SIMD to fill in.  Here block i+1's score MFMAs are
// issued in the same scheduling region as block i's softmax arithmetic (two score-register sets; the key-padding mask enters as
// the accumulator's initial value, so the region has no branches), interleaved one MFMA + one fragment read to a handful of
// VALU / transcendental instructions by sched_group_barrier.  Arithmetic per query is unchanged: same bits as attention_kernel.
// ------------------------------------------------------------------------------------------------
template <int MAXKB>
__global__ __launch_bounds__(256, 2) void attention_pipe_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int T, int H,
                                                               int ld_qkv_, int ld_ctx_, int k_off, int v_off, SeqLayout sl) {
  __shared__ __attribute__((aligned(16))) char smem[2 * MAXKB * 16 * 128];
  char* Ks = smem;
  char* Vs = smem + MAXKB * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq = blockIdx.x / H, h = blockIdx.x % H;
  const size_t row0 = (size_t)(seq / sl.inner_count) * sl.outer_rows + (size_t)(seq % sl.inner_count) * sl.inner_rows;
  const size_t ld_qkv = (size_t)ld_qkv_ * sl.row_step, ld_ctx = (size_t)ld_ctx_ * sl.row_step;
  const bf16_t* base = qkv + row0 * ld_qkv_ + h * 64;
  constexpr int nkc = MAXKB / 2;
  constexpr int tpad = MAXKB * 16;
  {
    constexpr int NIT = (MAXKB * 16 * 8 + 255) / 256;
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
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  const int nqb = (T + 15) >> 4;
  auto load_q = [&](int qb, bf16x8 (&dst)[2]) {
    int qrow = qb * 16 + fr;
    if (qrow >= T) qrow = T - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) dst[kk] = *(const bf16x8*)(base + (size_t)qrow * ld_qkv + kk * 32 + fq * 8);
  };
  // per-lane K fragment address of key block 0 (row fr); block kb adds kb * 2048
  const int koff0 = fr * 128 + ((fq ^ (fr & 7)) << 4), koff1 = fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
  const int tl = T - fq * 4;                  // key kb*16 + fq*4 + r is padding iff kb*16 + r >= tl
  // the score accumulators start at 0, or at -3e38 for keys >= T (their K rows are zero: the products add nothing)
  auto init_block = [&](int kb) -> f32x4 {
    f32x4 z;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = (kb >= MAXKB - 6 && kb * 16 + r >= tl) ? -3.0e38f : 0.f;
    return z;
  };
  auto scores = [&](const bf16x8 (&q)[2], f32x4 (&dst)[MAXKB]) {      // unpipelined (first block of a wave)
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb) {
      f32x4 a = init_block(kb);
      a = mfma_op16(*(const bf16x8*)(Ks + kb * 2048 + koff0), q[0], a);
      dst[kb] = mfma_op16(*(const bf16x8*)(Ks + kb * 2048 + koff1), q[1], a);
    }
  };

  bf16x8 qn[2];
  f32x4 st[MAXKB], stn[MAXKB];
  if (wave < nqb) {
    bf16x8 qf[2];
    load_q(wave, qf);
    if (wave + 4 < nqb) load_q(wave + 4, qn);
    scores(qf, st);
  }
  const f32x2 l2e = {1.44269504088896341f, 1.44269504088896341f};
  for (int qb = wave; qb < nqb; qb += 4) {
    const bool has_next = qb + 4 < nqb;       // wave-uniform
    // ---- region A: S(next) on the matrix pipe + softmax(cur) on the VALU, one scheduling region ----
    float mx = -3.0e38f;
#pragma unroll
    for (int kb = 0; kb < MAXKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][r]);
    mx = rows4_max(mx);
    const float mneg1 = -mx * 1.44269504088896341f;
    const f32x2 mneg = {mneg1, mneg1};
    f32x2 sum2 = {0.f, 0.f};
    if (has_next) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) {
        // two MFMAs of the next block's key block kb ...
        f32x4 a = init_block(kb);
#if defined(PG_PIPE_ABL) && PG_PIPE_ABL == 1      /* timing ablation (probe builds): one K fragment pair for all key blocks */
        const bf16x8 k0 = *(const bf16x8*)(Ks + koff0), k1 = *(const bf16x8*)(Ks + koff1);
#else
        const bf16x8 k0 = *(const bf16x8*)(Ks + kb * 2048 + koff0), k1 = *(const bf16x8*)(Ks + kb * 2048 + koff1);
#endif
        a = mfma_op16(k0, qn[0], a);
        stn[kb] = mfma_op16(k1, qn[1], a);
        // ... beside the exponentials of this block's key block kb
        const f32x2 x0 = __builtin_elementwise_fma((f32x2){st[kb][0], st[kb][1]}, l2e, mneg);
        const f32x2 x1 = __builtin_elementwise_fma((f32x2){st[kb][2], st[kb][3]}, l2e, mneg);
#if defined(PG_PIPE_ABL) && PG_PIPE_ABL == 3      /* timing ablation: no transcendentals */
        const f32x2 ea = x0, eb = x1;
#else
        const f32x2 ea = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
        const f32x2 eb = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
#endif
        st[kb] = (f32x4){ea[0], ea[1], eb[0], eb[1]};
        sum2 += ea;
        sum2 += eb;
      }
      // interleave: per key block 2 MFMAs beside ~10 VALU + 4 transcendentals, the two K fragment reads of the NEXT key block
      // issued ahead of them
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // DS read: key block 0
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // DS read (key block kb + 1)
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);    // VALU
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);    // TRANS
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // VALU
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);    // TRANS
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // VALU
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) {
        const f32x2 x0 = __builtin_elementwise_fma((f32x2){st[kb][0], st[kb][1]}, l2e, mneg);
        const f32x2 x1 = __builtin_elementwise_fma((f32x2){st[kb][2], st[kb][3]}, l2e, mneg);
        const f32x2 ea = {__builtin_amdgcn_exp2f(x0[0]), __builtin_amdgcn_exp2f(x0[1])};
        const f32x2 eb = {__builtin_amdgcn_exp2f(x1[0]), __builtin_amdgcn_exp2f(x1[1])};
        st[kb] = (f32x4){ea[0], ea[1], eb[0], eb[1]};
        sum2 += ea;
        sum2 += eb;
      }
    }
    float sum = sum2[0] + sum2[1];
    sum = rows4_sum(sum);
    const float inv = 1.0f / sum;
    if (qb + 8 < nqb) load_q(qb + 8, qn);     // the block after next: its fragment is needed by the NEXT iteration's region A

    // ---- region B: O^T = V^T P^T (as attention_kernel) ----
    f32x4 o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      union VF { bf16x8 v; uint2 h[2]; };
      VF vbuf[3][4];
      auto load_v = [&](int c, VF (&dst)[4]) {
#if defined(PG_PIPE_ABL) && PG_PIPE_ABL == 2      /* timing ablation: the V^T fragments of chunk 0 for every chunk */
        c = 0;
#endif
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int krow = (2 * c + hh) * 16 + fq * 4 + (fr >> 2);
            const int dcol = db * 16 + (fr & 3) * 4;
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
    if (has_next) {
#pragma unroll
      for (int kb = 0; kb < MAXKB; ++kb) st[kb] = stn[kb];
    }
  }
}

PG_OPS_END

namespace pg {
int fail(int code, const std::string& msg) { fprintf(stderr, "%s\n", msg.c_str()); return code; }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 258, H = 20, d = H * 64;
  const size_t M = (size_t)B * T;
  pg::bf16_t *qkv, *c0, *c1;
  (void)hipMalloc(&qkv, M * 3 * d * 2);
  (void)hipMalloc(&c0, M * d * 2);
  (void)hipMalloc(&c1, M * d * 2);
  std::vector<pg::bf16_t> h(M * 3 * d);
  unsigned st = 1;
  for (size_t i = 0; i < h.size(); ++i) {
    st = st * 1664525u + 1013904223u;
    const float u = ((st >> 8) * (1.0f / 8388608.0f) - 1.0f);
    h[i] = pg::f32_to_bf16(u * ((i % (3 * d)) < (size_t)d ? 0.6f : 1.5f));      // q pre-scaled, k / v wider
  }
  (void)hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const pg::SeqLayout sl = {1, T, 0, 1};
  dim3 grid(B * H), block(256);
  auto run0 = [&]() { hipLaunchKernelGGL((pg::attention_kernel<18, false>), grid, block, 0, nullptr, qkv, c0, T, H, 3 * d, d, d, 2 * d, sl, nullptr, -1, nullptr); };
  auto run1 = [&]() { hipLaunchKernelGGL((pg::attention_pipe_kernel<18>), grid, block, 0, nullptr, qkv, c1, T, H, 3 * d, d, d, 2 * d, sl); };
  (void)hipMemset(c0, 0xff, M * d * 2);
  (void)hipMemset(c1, 0xee, M * d * 2);
  run0(); run1();
  (void)hipDeviceSynchronize();
  std::vector<pg::bf16_t> a(M * d), b(M * d);
  (void)hipMemcpy(a.data(), c0, a.size() * 2, hipMemcpyDeviceToHost);
  (void)hipMemcpy(b.data(), c1, b.size() * 2, hipMemcpyDeviceToHost);
  size_t ndiff = 0; double maxd = 0;
  for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { ++ndiff; const double dd = fabs(pg::bf16_to_f32(a[i]) - pg::bf16_to_f32(b[i])); if (dd > maxd) maxd = dd; }
  printf("outputs: %zu of %zu values differ (max |diff| %.3e)  hipGetLastError=%s\n", ndiff, a.size(), maxd, hipGetErrorString(hipGetLastError()));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int round = 0; round < 4; ++round)
    for (int which = 0; which < 2; ++which) {
      for (int i = 0; i < 5; ++i) which ? run1() : run0();
      (void)hipEventRecord(e0);
      for (int i = 0; i < 40; ++i) which ? run1() : run0();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("round %d %s: %.1f us per launch\n", round, which ? "pipe    " : "baseline", ms * 1e3 / 40);
    }
  return ndiff ? 1 : 0;
}
