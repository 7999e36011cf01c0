// The transformer trunk of a SINGLE short chain (<= 32 token rows: BASELINE config 1, one 25-residue chain) as ONE persistent
// launch: every layer's LayerNorm -> QKV -> attention -> out-proj -> LayerNorm -> fc1 + GELU -> fc2 of
// fair-esm's ProteinBertModel (reached through `self.model.model(batch)["logits"]`, /root/reference/src/pgen/esm_sampler.py:223).
//
// Why.  With 27 token rows a layer is 39 MB of weights streamed once and next to no arithmetic: as separate launches (round 3:
// gemm_ln_skinny, attention, skinny out-proj, gemm_ln_skinny, split-K fc2 + reduction = 6 launches per layer, 52 us) every
// launch pays its own dispatch ramp, kernel-argument fetch and -- the dominant term -- a chain of dependent memory round trips that
// starts only when the previous launch has drained: weights, activations, statistics, output.  Here one workgroup sits on every CU
// for the whole trunk.  The phases of a layer are separated by device-wide barriers (a monotonic counter in device memory,
// agent-scope release / acquire: buffer_wbl2 / buffer_inv sc1 across the 8 XCDs' L2s), and each workgroup issues the loads of its
// NEXT phase's weight tile between arriving at a barrier and waiting on it -- weights do not depend on activations, so their HBM
// latency is hidden behind the barrier instead of following it.
//
// Arithmetic.  Every phase is the statement-for-statement arithmetic of the kernel it replaces (gemm_ln_skinny_kernel,
// gemm_bf16_skinny_kernel<.., 8 waves>, attention_kernel<2>, the 4-way split-K fc2 + splitk_reduce_kernel): the same K split over
// the 8 waves, the same reduction orders, the same softmax.  The logits are BIT-IDENTICAL with the multi-launch path
// (PGIBBS_CHAIN_TRUNK=0); tests/test_gpu_chain_trunk.py compares the two.
//
// Work split (d = 1280): QKV 240 units of 16 features, out-proj 80 units, fc1 160 units of 2 x 16 features, fc2 160 units
// (40 feature pairs x 4 K-splits; the workgroup that completes a pair's fourth partial adds the four, in split order, to the
// residual stream -- "last arriver reduces", deterministic); attention one wave per (chain, head, 16-query block).
//
// A barrier that is not reached within 50 ms (two persistent grids sharing one GPU could starve each other) sets an error word
// instead of hanging the device; the engine reports it (PG_ERR_HIP) after the call.
#include "gemm_epilogue.h"
#include "kernels.h"

PG_OPS_BEGIN

typedef short ct_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(2))) float ct_f32x2;

namespace {

constexpr int CT_NW = 8;                       // waves per workgroup; wave w owns K slice [w * K/8, (w+1) * K/8) of every product

template <int NKS, int NB> struct CtW { bf16x8 w[NKS][NB]; };

struct CtShared {
  float red[CT_NW - 1][2][2][64][4];           // cross-wave partial sums (MT <= 2, NB <= 2)
  float stat[2][CT_NW][2][16];                 // LayerNorm row sums per wave
  __attribute__((aligned(16))) char vs[CT_NW][32 * 128];   // attention: one V tile (32 keys x 64 d) per wave
  int flag;
};

// ---- device-wide barrier, split in two so that the caller can put independent loads in between ----
__device__ __forceinline__ void ct_arrive(unsigned* sync) {
  __syncthreads();                             // every wave's stores of the phase have been acknowledged (s_waitcnt vmcnt(0))
  if (threadIdx.x == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ct_wait(unsigned* sync, unsigned target, bool& dead, unsigned* err) {
  if (threadIdx.x == 0) {
    if (!dead) {
      const long long t0 = wall_clock64();     // 100 MHz
      while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 5000000) {   // 50 ms: flag the failure, stop waiting at every later barrier too
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          dead = true;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // one invalidate after the wait, not one per poll
  }
  __builtin_amdgcn_s_barrier();                // raw: the prefetched weight loads stay in flight across it
}

// the thread index behind an empty asm: per-lane addresses derived from it are recomputed where they are used instead of being
// hoisted out of the layer loop (dozens of loop-invariant 64-bit addresses held across all phases spilled to scratch)
__device__ __forceinline__ int ct_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// every pointer here comes out of a struct (the kernel's argument block, the layer table) and is generic to the compiler: go
// through an explicit global address space so that the accesses are global_load / global_store, not flat_*
typedef unsigned ct_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ct_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T ct_ldg(const void* p) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8, "16- or 8-byte accesses");
  if constexpr (sizeof(T) == 16) return __builtin_bit_cast(T, *(const __attribute__((address_space(1))) ct_u32x4*)p);
  else return __builtin_bit_cast(T, *(const __attribute__((address_space(1))) ct_u32x2*)p);
}
template <typename T> __device__ __forceinline__ void ct_stg(void* p, T v) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8, "16- or 8-byte accesses");
  if constexpr (sizeof(T) == 16) *(__attribute__((address_space(1))) ct_u32x4*)p = __builtin_bit_cast(ct_u32x4, v);
  else *(__attribute__((address_space(1))) ct_u32x2*)p = __builtin_bit_cast(ct_u32x2, v);
}
#define CT_LDG(T, p) ct_ldg<T>(p)

template <int NKS, int NB>
__device__ __forceinline__ void ct_load_w(CtW<NKS, NB>& wf, const bf16_t* __restrict__ W, int ldw, int n0, int koff) {
  const int tid = ct_tid();
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const bf16_t* wp = W + (size_t)(n0 + fr) * ldw + koff + wave * (NKS * 32) + fq * 8;
#pragma unroll
  for (int u = 0; u < NKS; ++u)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wf.w[u][nb] = CT_LDG(bf16x8, wp + (size_t)nb * 16 * ldw + u * 32);
}

// out[MT*16][n0 .. n0 + 16 NB) = LayerNorm(x; gamma, beta) . W^T + bias (+ GELU), bf16.  The body of gemm_ln_skinny_kernel.
template <int MT, int NKS, int NB, int EPI>
__device__ __forceinline__ void ct_unit_ln(CtShared& sm, const float* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float eps, const CtW<NKS, NB>& wf,
                                           const float* __restrict__ bias, bf16_t* __restrict__ out, int ldo, int n0) {
  constexpr int NW = CT_NW;
  const int tid = ct_tid();
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int kq = NKS * 32;
  const int K = NW * kq;
  const int k0 = wave * kq + fq * 8;
  const float* xp = X + (size_t)fr * ldx + k0;
  float4 xa[NKS][MT][2];
#pragma unroll
  for (int u = 0; u < NKS; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      xa[u][t][0] = CT_LDG(float4, xp + (size_t)t * 16 * ldx + u * 32);
      xa[u][t][1] = CT_LDG(float4, xp + (size_t)t * 16 * ldx + u * 32 + 4);
    }
  const float inv_k = 1.0f / (float)K;
  float mean[MT], rstd[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NKS; ++u)
      s += ((xa[u][t][0].x + xa[u][t][0].y) + (xa[u][t][0].z + xa[u][t][0].w)) + ((xa[u][t][1].x + xa[u][t][1].y) + (xa[u][t][1].z + xa[u][t][1].w));
    s = rows4_sum(s);
    if (fq == 0) sm.stat[0][wave][t][fr] = s;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) s += sm.stat[0][w2][t][fr];
    mean[t] = s * inv_k;
  }
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
    if (fq == 0) sm.stat[1][wave][t][fr] = q;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    float q = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) q += sm.stat[1][w2][t][fr];
    rstd[t] = 1.0f / sqrtf(q * inv_k + eps);
  }
  f32x4 acc[MT][NB];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
    const float4 g0 = CT_LDG(float4, gamma + k0 + u * 32), g1 = CT_LDG(float4, gamma + k0 + u * 32 + 4);
    const float4 b0 = CT_LDG(float4, beta + k0 + u * 32), b1 = CT_LDG(float4, beta + k0 + u * 32 + 4);
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
      for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma_op16(wf.w[u][nb], __builtin_bit_cast(bf16x8, pk), acc[t][nb]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) *(f32x4*)&sm.red[wave - 1][t][nb][lane][0] = acc[t][nb];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const float4 b4 = CT_LDG(float4, bias + n0 + nb * 16 + fq * 4);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f32x4 v = acc[t][nb];
#pragma unroll
      for (int w2 = 0; w2 < NW - 1; ++w2) {
        const f32x4 o = *(const f32x4*)&sm.red[w2][t][nb][lane][0];
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
      }
      const float v0 = v[0] + b4.x, v1 = v[1] + b4.y, v2 = v[2] + b4.z, v3 = v[3] + b4.w;
      uint2 p;
      p.x = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v0, v1) : pack_op2(v0, v1);
      p.y = EPI == EPI_BF16_GELU ? gelu_bf16out_pack2(v2, v3) : pack_op2(v2, v3);
      ct_stg<uint2>(out + (size_t)(t * 16 + fr) * ldo + n0 + nb * 16 + fq * 4, p);
    }
  }
}

// out[MT*16][n0 .. n0 + 16 NB) (+)= A[MT*16][koff .. koff + 256 NKS) . W^T: EPI_F32_RESID adds bias and the product to the fp32
// residual stream in place, EPI_F32_PARTIAL stores the bare product.  The arithmetic of gemm_bf16_skinny_kernel<MT, EPI, 8>.
template <int MT, int NKS, int NB, int EPI>
__device__ __forceinline__ void ct_unit_bf16(CtShared& sm, const bf16_t* __restrict__ A, int lda, int koff, const CtW<NKS, NB>& wf,
                                             const float* __restrict__ bias, float* __restrict__ out, int ldo, int n0) {
  constexpr int NW = CT_NW;
  const int tid = ct_tid();
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const bf16_t* xp = A + (size_t)fr * lda + koff + wave * (NKS * 32) + fq * 8;
  bf16x8 xf[NKS][MT];
#pragma unroll
  for (int u = 0; u < NKS; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) xf[u][t] = CT_LDG(bf16x8, xp + (size_t)t * 16 * lda + u * 32);
  f32x4 acc[MT][NB];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NKS; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma_op16(wf.w[u][nb], xf[u][t], acc[t][nb]);
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) *(f32x4*)&sm.red[wave - 1][t][nb][lane][0] = acc[t][nb];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI != EPI_F32_PARTIAL) b4 = CT_LDG(float4, bias + n0 + nb * 16 + fq * 4);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      f32x4 v = acc[t][nb];
#pragma unroll
      for (int w2 = 0; w2 < NW - 1; ++w2) {
        const f32x4 o = *(const f32x4*)&sm.red[w2][t][nb][lane][0];
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
      }
      const float v0 = v[0] + b4.x, v1 = v[1] + b4.y, v2 = v[2] + b4.z, v3 = v[3] + b4.w;
      float* dst = out + (size_t)(t * 16 + fr) * ldo + n0 + nb * 16 + fq * 4;
      if (EPI == EPI_F32_RESID) {
        float4 r = CT_LDG(float4, dst);
        r.x += v0; r.y += v1; r.z += v2; r.w += v3;
        ct_stg<float4>(dst, r);
      } else {
        ct_stg<float4>(dst, make_float4(v0, v1, v2, v3));
      }
    }
  }
}

// One wave: 16 queries of one (chain, head) against the chain's T <= 32 keys.  The arithmetic of attention_kernel<2, false>;
// K fragments come straight from global memory (zero rows past T), V goes through the wave's own 4 KB of LDS for the
// transposing read.
__device__ __forceinline__ void ct_attention_unit(char* __restrict__ Vs, const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                  int T, int d, int seq, int h, int qb) {
  const int lane = ct_tid() & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const size_t ld_qkv = (size_t)3 * d, ld_ctx = (size_t)d;
  const size_t row0 = (size_t)seq * T;
  const bf16_t* base = qkv + row0 * ld_qkv + h * 64;
  const int k_off = d, v_off = 2 * d;
  {
    uint4 vreg[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = lane + it * 64;
      const int row = i >> 3, c = i & 7;
      vreg[it] = make_uint4(0, 0, 0, 0);
      if (row < T) vreg[it] = CT_LDG(uint4, base + (size_t)row * ld_qkv + v_off + c * 8);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = lane + it * 64;
      const int row = i >> 3, c = i & 7;
      *(uint4*)(Vs + row * 128 + ((c ^ (row & 7)) << 4)) = vreg[it];
    }
  }
  asm volatile("" ::: "memory");               // the V tile is the wave's own: LDS executes a wave's accesses in order, no barrier
  bf16x8 qf[2], kf[2][2];
  {
    int qrow = qb * 16 + fr;
    if (qrow >= T) qrow = T - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = CT_LDG(bf16x8, base + (size_t)qrow * ld_qkv + kk * 32 + fq * 8);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int krow = u * 16 + fr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 z = make_uint4(0, 0, 0, 0);
        if (krow < T) z = CT_LDG(uint4, base + (size_t)krow * ld_qkv + k_off + (kk * 4 + fq) * 8);
        kf[u][kk] = __builtin_bit_cast(bf16x8, z);
      }
    }
  }
  f32x4 st[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) st[u] = mfma_op16(kf[u][0], qf[0], (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
  for (int u = 0; u < 2; ++u) st[u] = mfma_op16(kf[u][1], qf[1], st[u]);
  float mx = -3.0e38f;
  const int tl = T - fq * 4;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
    if ((kb + 1) * 16 > T) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (kb * 16 + r >= tl) st[kb][r] = -3.0e38f;
    }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][r]);
  mx = rows4_max(mx);
  const ct_f32x2 l2e = {1.44269504088896341f, 1.44269504088896341f};
  const float mneg1 = -mx * 1.44269504088896341f;
  const ct_f32x2 mneg = {mneg1, mneg1};
  ct_f32x2 sum2 = {0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const ct_f32x2 a = __builtin_elementwise_fma((ct_f32x2){st[kb][0], st[kb][1]}, l2e, mneg);
    const ct_f32x2 b = __builtin_elementwise_fma((ct_f32x2){st[kb][2], st[kb][3]}, l2e, mneg);
    const ct_f32x2 ea = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const ct_f32x2 eb = {__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
    st[kb] = (f32x4){ea[0], ea[1], eb[0], eb[1]};
    sum2 += ea;
    sum2 += eb;
  }
  float sum = sum2[0] + sum2[1];
  sum = rows4_sum(sum);
  const float inv = 1.0f / sum;
  f32x4 o[4];
  union VF { bf16x8 v; uint2 h[2]; };
  VF vb[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int krow = hh * 16 + fq * 4 + (fr >> 2);
      const int dcol = db * 16 + (fr & 3) * 4;
      const char* a = Vs + krow * 128 + (((dcol >> 3) ^ (krow & 7)) << 4) + ((dcol >> 2) & 1) * 8;
      const ct_v4s t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ct_v4s __attribute__((address_space(3)))*)(
          (__attribute__((address_space(3))) char*)a));
      vb[db].h[hh] = __builtin_bit_cast(uint2, t);
    }
  }
  union { bf16x8 v; uint32_t u[4]; } pf;
  pf.u[0] = pack_op2(st[0][0], st[0][1]);
  pf.u[1] = pack_op2(st[0][2], st[0][3]);
  pf.u[2] = pack_op2(st[1][0], st[1][1]);
  pf.u[3] = pack_op2(st[1][2], st[1][3]);
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = mfma_op16(vb[db].v, pf.v, (f32x4){0.f, 0.f, 0.f, 0.f});
  const int q = qb * 16 + fr;
  if (q < T) {
    bf16_t* dst = ctx + row0 * ld_ctx + (size_t)q * ld_ctx + h * 64 + fq * 4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      uint2 p;
      p.x = pack_op2(o[db][0] * inv, o[db][1] * inv);
      p.y = pack_op2(o[db][2] * inv, o[db][3] * inv);
      ct_stg<uint2>(dst + db * 16, p);
    }
  }
}

}  // namespace

// MT = 16-row tiles of the token rows (1 or 2), NKS = d_model / 256 (d_ffn = 4 d_model, heads of 64)
template <int MT, int NKS>
__global__ __launch_bounds__(512) void chain_trunk_kernel(PgChainTrunkArgs a) {
  constexpr int d = NKS * 256, f = 4 * d;
  constexpr int nQ = 3 * d / 16, nO = d / 16, nF1 = f / 32, nPair = d / 32, nF2 = nPair * 4;
  __shared__ CtShared sm;
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  unsigned* sync = a.sync;
  unsigned* pair_cnt = a.sync + 16;
  unsigned phase = 0;
  bool dead = false;
  const int H = d / 64;
  const int nqb = (a.T + 15) >> 4;
  const int n_att = a.B * H * nqb;
  float* X = a.x;
  float* part = a.part;
  const long part_stride = (long)MT * 16 * d;

  CtW<NKS, 1> w1;
  CtW<NKS, 2> w2;
  if (b < nQ) ct_load_w(w1, a.layers[0].qkv_w, d, b * 16, 0);
  for (int l = 0; l < a.n_layers; ++l) {
    const PgChainLayerW lw = a.layers[l];
    const bool last = l + 1 == a.n_layers;
    // ---- x -> LayerNorm -> q | k | v
    for (int u = b; u < nQ; u += G) {
      if (u != b) { __syncthreads(); ct_load_w(w1, lw.qkv_w, d, u * 16, 0); }
      ct_unit_ln<MT, NKS, 1, EPI_BF16>(sm, X, d, lw.ln1_g, lw.ln1_b, a.eps, w1, lw.qkv_b, a.qkv, 3 * d, u * 16);
    }
    ct_arrive(sync);
    const bool tail = !(last && a.partial_last);          // out-proj and the feed-forward block of this layer run here
    if (tail && b < nO) ct_load_w(w1, lw.out_w, d, b * 16, 0);
    ct_wait(sync, ++phase * G, dead, a.err);
    // ---- attention: wave-units spread over the workgroups
    for (int wu = b + G * wave; wu < n_att; wu += G * CT_NW) {
      const int qb = wu % nqb, sh = wu / nqb;
      ct_attention_unit(sm.vs[wave], a.qkv, a.ctx, a.T, d, sh / H, sh % H, qb);
    }
    ct_arrive(sync);
    ct_wait(sync, ++phase * G, dead, a.err);
    if (!tail) break;
    // ---- x += out_proj(ctx)
    for (int u = b; u < nO; u += G) {
      if (u != b) { __syncthreads(); ct_load_w(w1, lw.out_w, d, u * 16, 0); }
      ct_unit_bf16<MT, NKS, 1, EPI_F32_RESID>(sm, a.ctx, d, 0, w1, lw.out_b, X, d, u * 16);
    }
    ct_arrive(sync);
    if (b < nF1) ct_load_w(w2, lw.fc1_w, d, b * 32, 0);
    ct_wait(sync, ++phase * G, dead, a.err);
    // ---- ffn = gelu(fc1(LayerNorm(x)))
    for (int u = b; u < nF1; u += G) {
      if (u != b) { __syncthreads(); ct_load_w(w2, lw.fc1_w, d, u * 32, 0); }
      ct_unit_ln<MT, NKS, 2, EPI_BF16_GELU>(sm, X, d, lw.ln2_g, lw.ln2_b, a.eps, w2, lw.fc1_b, a.ffn, f, u * 32);
    }
    ct_arrive(sync);
    if (b < nF2) ct_load_w(w2, lw.fc2_w, f, (b % nPair) * 32, (b / nPair) * d);
    ct_wait(sync, ++phase * G, dead, a.err);
    // ---- x += fc2(ffn): four K-splits per feature pair; the last one to finish adds them in split order
    for (int u = b; u < nF2; u += G) {
      const int pair = u % nPair, split = u / nPair;
      if (u != b) { __syncthreads(); ct_load_w(w2, lw.fc2_w, f, pair * 32, split * d); }
      ct_unit_bf16<MT, NKS, 2, EPI_F32_PARTIAL>(sm, a.ffn, f, split * d, w2, nullptr, part + (size_t)split * part_stride, d, pair * 32);
      __syncthreads();                                     // wave 0's partial stores are acknowledged
      if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(pair_cnt + pair, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        sm.flag = old == 3u;
        if (old == 3u) __hip_atomic_store(pair_cnt + pair, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (sm.flag && tid < MT * 16 * 8) {                  // splitk_reduce_kernel's arithmetic on this pair's 32 features
        const int m = tid >> 3, c = tid & 7;
        const size_t o = (size_t)m * d + pair * 32 + c * 4;
        float4 acc = CT_LDG(float4, lw.fc2_b + pair * 32 + c * 4);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const float4 p = CT_LDG(float4, part + (size_t)sidx * part_stride + o);
          acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        float4 r = CT_LDG(float4, X + o);
        r.x += acc.x; r.y += acc.y; r.z += acc.z; r.w += acc.w;
        ct_stg<float4>(X + o, r);
      }
    }
    ct_arrive(sync);
    if (!last && b < nQ) ct_load_w(w1, a.layers[l + 1].qkv_w, d, b * 16, 0);
    ct_wait(sync, ++phase * G, dead, a.err);
  }
  // every workgroup is past its last wait once the exit count is full: the last one out rearms the counters for the next launch
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)G - 1u) {
      __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// may the persistent trunk take this forward?  M = padded token rows (16 or 32)
bool chain_trunk_ok(int M, int d_model, int d_ffn, int n_heads) {
  static const int on = [] { const char* e = getenv("PGIBBS_CHAIN_TRUNK"); return e ? atoi(e) : 1; }();
  return on && (M == 16 || M == 32) && d_model % 256 == 0 && d_model / 256 >= 1 && d_model / 256 <= 5 && d_ffn == 4 * d_model &&
         n_heads * 64 == d_model;
}
size_t chain_trunk_sync_bytes() { return 4096; }           // barrier counter, error word, exit counter, pair counters (zeroed once)
size_t chain_trunk_part_bytes(int M, int d_model) { return (size_t)4 * M * d_model * 4; }

int launch_chain_trunk(hipStream_t s, const PgChainTrunkArgs& a, int M, int d_model) {
  static const int n_cu = [] { hipDeviceProp_t p; int dv = 0; (void)hipGetDevice(&dv); return hipGetDeviceProperties(&p, dv) == hipSuccess ? p.multiProcessorCount : 256; }();
  static const int g_env = [] { const char* e = getenv("PGIBBS_CHAIN_TRUNK_GRID"); return e ? atoi(e) : 0; }();
  // one workgroup per CU: all of them must be resident at once (the barriers spin), and 512 threads + ~65 KB of LDS fit any CU
  const int G = g_env > 0 && g_env <= n_cu ? g_env : n_cu;
  if (a.n_layers <= 0) return 0;
  if (a.B * a.T > M || a.T > 32 || a.T < 1) return fail(1, "chain_trunk: shape");
  dim3 grid(G), block(512);
#define PG_CT(MTV, NK) hipLaunchKernelGGL((chain_trunk_kernel<MTV, NK>), grid, block, 0, s, a)
#define PG_CT_NK(MTV)                          \
  switch (d_model / 256) {                     \
    case 1: PG_CT(MTV, 1); break;              \
    case 2: PG_CT(MTV, 2); break;              \
    case 3: PG_CT(MTV, 3); break;              \
    case 4: PG_CT(MTV, 4); break;              \
    default: PG_CT(MTV, 5); break;             \
  }
  if (M == 16) { PG_CT_NK(1) } else { PG_CT_NK(2) }
#undef PG_CT_NK
#undef PG_CT
  PG_HIP(hipGetLastError());
  return 0;
}

PG_OPS_END
