// The transformer trunk of a SINGLE short chain (<= 32 token rows: BASELINE config 1, one 25-residue chain) as ONE persistent
// launch: every layer's LayerNorm -> QKV -> attention -> out-proj -> LayerNorm -> fc1 + GELU -> fc2 of
// fair-esm's ProteinBertModel (reached through `self.model.model(batch)["logits"]`, /root/reference/src/pgen/esm_sampler.py:223).
//
// Why.  With 27 token rows a layer is 39 MB of weights streamed once and next to no arithmetic: as separate launches (round 3:
// gemm_ln_skinny, attention, skinny out-proj, gemm_ln_skinny, split-K fc2 + reduction = 6 launches per layer, 52 us) every
// launch boundary costs ~5 us of drain / cache maintenance / dispatch, and every launch starts its chain of dependent memory
// round trips (weights, activations, statistics, output) from cold.  Here one workgroup sits on every CU for the whole trunk, the
// phases of a layer are separated by device-wide barriers, and each workgroup issues the loads of its NEXT phase's weight tile
// between arriving at a barrier and waiting on it -- weights do not depend on activations, so their HBM latency hides behind the
// barrier instead of following it.
//
// The barrier and the data path (measured: tools/probes/grid_barrier_bench.hip, profiles/r04_chain_trunk_barrier_probe.txt).
// The textbook form -- one counter, agent-scope release / acquire -- costs 10 us per barrier on the 8-XCD part: every one of the
// 256 workgroups executes buffer_wbl2 + buffer_inv (a walk of its XCD's 4 MB L2), 32 of them per L2, and the phases' loads queue
// behind the walks (17-19 us per phase).  So there is NO cache maintenance here:
//   * everything one phase writes and a later phase reads (x, q|k|v, context, FFN rows, fc2 partials) moves with sc1 = DEVICE-SCOPE
//     buffer loads and stores: the ISA's own contract for data shared across the device -- written through to, and fetched from,
//     the memory side, whatever the XCDs' L2s hold (weights, LayerNorm parameters and biases are read-only: ordinary cached loads).
//     (Reading the rows through the L2s instead -- every buffer at a fresh address per layer, so that no L2 could hold a stale line --
//     was built and measured: same time.  A CU's load path, not the memory side, bounds the row reads: see ct_unit_ln.);
//   * arriving = s_waitcnt vmcnt(0) (the stores are acknowledged) + ONE relaxed store of the epoch into the workgroup's own slot;
//     the last workgroup of the grid -- idle in every phase: no phase has more than 240 units -- polls the 256 slots (one dwordx4
//     per lane) and publishes the epoch in 8 flags; every workgroup polls flag (blockIdx % 8).  No read-modify-write on a shared
//     word, no serialisation: 2.2 us per barrier.
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
// All workgroups must be resident at once (the barriers spin).  Alone on the device they are: one 8-wave workgroup per CU.  Two
// persistent grids sharing one GPU -- two processes sampling single chains -- can each hold half of the CUs and starve each other
// (measured: 21 timeouts in 150 calls), so a barrier that does not open within 50 ms sets a host-visible error word instead of
// hanging the device, every later wait of that workgroup is skipped, and the engine (Engine::chain_check) switches to the per-layer
// launches for good and re-runs the call from the caller's intact inputs.
#include <algorithm>

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
  __attribute__((aligned(16))) char vs[CT_NW][32 * 144];   // per wave: attention's V tile (32 keys x 64 d); operand-row staging (ct_stage_*)
  int flag;
};

// Phase timestamps for tools/probes/chain_trunk_phases.hip (compiled out of the library): thread 0 of every workgroup records the
// 100 MHz clock when its workgroup has finished a phase's work, when it has published its arrival, and when the barrier opened.
#ifdef PG_CT_TIMING
__device__ long long* pg_ct_stamps;            // [workgroup][1024]
__device__ __forceinline__ void ct_stamp(int& si) {
  if (threadIdx.x == 0 && si < 1024) pg_ct_stamps[(size_t)blockIdx.x * 1024 + si] = wall_clock64();
  ++si;
}
#define CT_STAMP() ct_stamp(si)
#define CT_STAMP_DECL int si = 0;
#else
#define CT_STAMP()
#define CT_STAMP_DECL
#endif

// ---- device-wide barrier, split in two so that the caller can put independent loads in between.  Words of the sync block:
// [2] exit counter, [16 .. 80) fc2 pair counters, [192 + 16 g] flag of group g (8 groups, one 64-byte line each),
// [512 + b] slot of workgroup b (16 slots per line).  `epoch` counts the barriers of this launch from 1.
#define CT_LD_RLX(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CT_ST_RLX(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
constexpr int CT_FLAGS = 192, CT_SLOTS = 512, CT_MAX_GRID = 512;
constexpr long long CT_TIMEOUT = 5000000;      // 50 ms of the 100 MHz clock

#ifdef PG_CT_TIMING
#define ct_arrive(sync, epoch) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); CT_STAMP(); if (threadIdx.x == 0) CT_ST_RLX((sync) + CT_SLOTS + blockIdx.x, (epoch)); CT_STAMP(); } while (0)
#else
__device__ __forceinline__ void ct_arrive(unsigned* sync, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's sc1 stores of the phase have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) CT_ST_RLX(sync + CT_SLOTS + blockIdx.x, epoch);
}
#endif
__device__ __forceinline__ void ct_wait(unsigned* sync, unsigned epoch, bool& dead, unsigned* err) {
  const int G = gridDim.x;
  if ((int)blockIdx.x == G - 1 && threadIdx.x < 64) {      // the aggregator wave: lane l watches slots 4l .. 4l+3 (+ 256, if any)
    const int l = threadIdx.x;
    const long long t0 = wall_clock64();
    bool expired = __any(dead);
    while (!expired) {
      bool ok = true;
      for (int w = 4 * l; w < G; w += 256) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (w + i < G) ok = ok && CT_LD_RLX(sync + CT_SLOTS + w + i) >= epoch;
      }
      if (__all(ok)) break;
      expired = __any(wall_clock64() - t0 > CT_TIMEOUT);
    }
    if (expired) dead = true;
    if (l < 8) CT_ST_RLX(sync + CT_FLAGS + 16 * l, epoch);  // open it even after a timeout: nobody may hang
  }
  if (threadIdx.x == 0) {
    const unsigned* flag = sync + CT_FLAGS + 16 * (blockIdx.x & 7);
    const long long t0 = wall_clock64();
    while (!dead && CT_LD_RLX(flag) < epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > CT_TIMEOUT) dead = true;
    }
    if (dead) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
#define CT_LDG(T, p) ct_ldg<T>(p)

// ---- inter-phase data: device-scope (sc1) buffer accesses.  aux bit 4 = sc1 in gfx940's cache policy (bit 0 = sc0, bit 1 = nt)
#ifndef CT_DEV_AUX
#define CT_DEV_AUX 16
#endif
constexpr int CT_SC1 = CT_DEV_AUX;
__device__ __forceinline__ rsrc_t ct_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000); }
template <typename T> __device__ __forceinline__ T ct_ld_dev(rsrc_t rs, int byte_off) {
  static_assert(sizeof(T) == 16, "16-byte accesses");
  return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, CT_SC1));
}
template <typename T> __device__ __forceinline__ void ct_st_dev(rsrc_t rs, int byte_off, T v) {
  static_assert(sizeof(T) == 16 || sizeof(T) == 8, "16- or 8-byte accesses");
  if constexpr (sizeof(T) == 16) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ct_u32x4, v), rs, byte_off, 0, CT_SC1);
  else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ct_u32x2, v), rs, byte_off, 0, CT_SC1);
}

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
__device__ __forceinline__ void ct_unit_ln(CtShared& sm, const float* X, int ldx, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float eps, const CtW<NKS, NB>& wf,
                                           const float* __restrict__ bias, bf16_t* out, int ldo, int n0) {
  constexpr int NW = CT_NW;
  const int tid = ct_tid();
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int kq = NKS * 32;
  const int K = NW * kq;
  const int k0 = wave * kq + fq * 8;
  const rsrc_t rx = ct_rsrc(X), ro = ct_rsrc(out);
  // The MFMA operand layout puts consecutive lanes on consecutive ROWS (lane = fq * 16 + fr holds row fr): loaded that way, the four
  // lanes of every quad hit four different cache lines and the texture addresser spends 64 clocks on a 1 KB load instead of 16
  // (measured: 4.6 us for the 138 KB of rows a workgroup reads, a quarter of the layer).  So the rows are loaded with consecutive
  // lanes on consecutive 16-byte pieces (8 lanes = one 128-byte line) and turned into the operand layout through the wave's own
  // 4.5 KB of LDS (rows padded to 144 bytes: two-way conflicts at most on either side); same values in the same registers.
  char* st = sm.vs[wave];
  const int xoff = ((lane >> 3) * ldx + wave * kq) * 4 + (lane & 7) * 16;
  float4 gx[NKS][MT * 2];
#pragma unroll
  for (int u = 0; u < NKS; ++u)
#pragma unroll
    for (int i = 0; i < MT * 2; ++i) {
#if defined(PG_CT_ABL) && PG_CT_ABL == 1       /* timing ablation (tools/probes/chain_trunk_phases.hip): no activation loads */
      gx[u][i] = make_float4((float)lane, 1.f, 2.f, (float)(u + i));
#else
      gx[u][i] = ct_ld_dev<float4>(rx, xoff + (i * 8 * ldx + u * 32) * 4);
#endif
    }
  float4 xa[NKS][MT][2];
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
#pragma unroll
    for (int i = 0; i < MT * 2; ++i) *(float4*)(st + (i * 8 + (lane >> 3)) * 144 + (lane & 7) * 16) = gx[u][i];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      xa[u][t][0] = *(const float4*)(st + (t * 16 + fr) * 144 + fq * 32);
      xa[u][t][1] = *(const float4*)(st + (t * 16 + fr) * 144 + fq * 32 + 16);
    }
    asm volatile("" ::: "memory");             // the next step's writes stay behind these reads (a wave's LDS accesses execute in order)
  }
  // LayerNorm parameters and bias: issued now (the row registers are free again), so that their round trip overlaps the statistics
  // instead of following them
  // (the first three k-steps' worth; the rest after the statistics: all five at once do not fit the register file next to
  // the prefetched weight tile)
  constexpr int NG1 = 0;
  float4 gb[NKS][4];
  auto load_gb = [&](int u) {
#if defined(PG_CT_ABL) && PG_CT_ABL == 2       /* timing ablation: no LayerNorm parameter loads */
    gb[u][0] = gb[u][1] = make_float4(1.f, 1.f, 1.f, 1.f);
    gb[u][2] = gb[u][3] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
    gb[u][0] = CT_LDG(float4, gamma + k0 + u * 32); gb[u][1] = CT_LDG(float4, gamma + k0 + u * 32 + 4);
    gb[u][2] = CT_LDG(float4, beta + k0 + u * 32); gb[u][3] = CT_LDG(float4, beta + k0 + u * 32 + 4);
#endif
  };
#pragma unroll
  for (int u = 0; u < NG1; ++u) load_gb(u);
  float4 bias4[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias4[nb] = CT_LDG(float4, bias + n0 + nb * 16 + fq * 4);
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
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = NG1; u < NKS; ++u) load_gb(u);
  f32x4 acc[MT][NB];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
    const float4 g0 = gb[u][0], g1 = gb[u][1], b0 = gb[u][2], b1 = gb[u][3];
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
    const float4 b4 = bias4[nb];
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
      ct_st_dev<uint2>(ro, ((t * 16 + fr) * ldo + n0 + nb * 16 + fq * 4) * 2, p);
    }
  }
}

// out[MT*16][n0 .. n0 + 16 NB) (+)= A[MT*16][koff .. koff + 256 NKS) . W^T: EPI_F32_RESID adds bias and the product to the fp32
// residual stream in place, EPI_F32_PARTIAL stores the bare product.  The arithmetic of gemm_bf16_skinny_kernel<MT, EPI, 8>.
template <int MT, int NKS, int NB, int EPI>
__device__ __forceinline__ void ct_unit_bf16(CtShared& sm, const bf16_t* A, int lda, int koff, const CtW<NKS, NB>& wf,
                                             const float* __restrict__ bias, float* out, int ldo, int n0) {
  constexpr int NW = CT_NW;
  const int tid = ct_tid();
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const rsrc_t ra = ct_rsrc(A), ro = ct_rsrc(out);
  // operand rows through the wave's LDS, as in ct_unit_ln: 4 lanes = the 64 bytes one row contributes to a 32-wide k-step
  char* st = sm.vs[wave];
  const int aoff = ((lane >> 2) * lda + koff + wave * (NKS * 32)) * 2 + (lane & 3) * 16;
  bf16x8 ga[NKS][MT];
#pragma unroll
  for (int u = 0; u < NKS; ++u)
#pragma unroll
    for (int t = 0; t < MT; ++t) ga[u][t] = ct_ld_dev<bf16x8>(ra, aoff + (t * 16 * lda + u * 32) * 2);
  bf16x8 xf[NKS][MT];
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
#pragma unroll
    for (int t = 0; t < MT; ++t) *(bf16x8*)(st + (t * 16 + (lane >> 2)) * 80 + (lane & 3) * 16) = ga[u][t];
#pragma unroll
    for (int t = 0; t < MT; ++t) xf[u][t] = *(const bf16x8*)(st + (t * 16 + fr) * 80 + fq * 16);
    asm volatile("" ::: "memory");
  }
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
      const int ooff = ((t * 16 + fr) * ldo + n0 + nb * 16 + fq * 4) * 4;
      if (EPI == EPI_F32_RESID) {
        float4 r = ct_ld_dev<float4>(ro, ooff);
        r.x += v0; r.y += v1; r.z += v2; r.w += v3;
        ct_st_dev<float4>(ro, ooff, r);
      } else {
        ct_st_dev<float4>(ro, ooff, make_float4(v0, v1, v2, v3));
      }
    }
  }
}

// One wave: 16 queries of one (chain, head) against the chain's T <= 32 keys.  The arithmetic of attention_kernel<2, false>;
// K fragments come straight from global memory (zero rows past T), V goes through the wave's own 4 KB of LDS for the
// transposing read.
__device__ __forceinline__ void ct_attention_unit(char* __restrict__ Vs, const bf16_t* qkv, bf16_t* ctx, int T, int d, int seq, int h,
                                                  int qb) {
  const int lane = ct_tid() & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int ld_qkv = 3 * d, ld_ctx = d;
  const int row0 = seq * T;
  const rsrc_t rq = ct_rsrc(qkv), rc = ct_rsrc(ctx);
  const int base = (row0 * ld_qkv + h * 64) * 2;           // byte offsets from here on
  const int k_off = d, v_off = 2 * d;
  {
    uint4 vreg[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = lane + it * 64;
      const int row = i >> 3, c = i & 7;
      vreg[it] = make_uint4(0, 0, 0, 0);
      if (row < T) vreg[it] = ct_ld_dev<uint4>(rq, base + (row * ld_qkv + v_off + c * 8) * 2);
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
    for (int kk = 0; kk < 2; ++kk) qf[kk] = ct_ld_dev<bf16x8>(rq, base + (qrow * ld_qkv + kk * 32 + fq * 8) * 2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int krow = u * 16 + fr;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        uint4 z = make_uint4(0, 0, 0, 0);
        if (krow < T) z = ct_ld_dev<uint4>(rq, base + (krow * ld_qkv + k_off + (kk * 4 + fq) * 8) * 2);
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
    const int dst = ((row0 + q) * ld_ctx + h * 64 + fq * 4) * 2;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      uint2 p;
      p.x = pack_op2(o[db][0] * inv, o[db][1] * inv);
      p.y = pack_op2(o[db][2] * inv, o[db][3] * inv);
      ct_st_dev<uint2>(rc, dst + db * 16 * 2, p);
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
  CT_STAMP_DECL
  bool dead = false;
  const int H = d / 64;
  const int nqb = (a.T + 15) >> 4;
  const int n_att = a.B * H * nqb;
  constexpr int Mrows = MT * 16;
  const long part_stride = (long)Mrows * d;
  float* X = a.x;
  float* part = a.part;

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
    ++phase; ct_arrive(sync, phase);
    const bool tail = !(last && a.partial_last);          // out-proj and the feed-forward block of this layer run here
    if (tail && b < nO) ct_load_w(w1, lw.out_w, d, b * 16, 0);
    ct_wait(sync, phase, dead, a.err); CT_STAMP();
    // ---- attention: wave-units spread over the workgroups
    for (int wu = b + G * wave; wu < n_att; wu += G * CT_NW) {
      const int qb = wu % nqb, sh = wu / nqb;
      ct_attention_unit(sm.vs[wave], a.qkv, a.ctx, a.T, d, sh / H, sh % H, qb);
    }
    ++phase; ct_arrive(sync, phase);
    ct_wait(sync, phase, dead, a.err); CT_STAMP();
    if (!tail) break;
    // ---- x += out_proj(ctx)
    for (int u = b; u < nO; u += G) {
      if (u != b) { __syncthreads(); ct_load_w(w1, lw.out_w, d, u * 16, 0); }
      ct_unit_bf16<MT, NKS, 1, EPI_F32_RESID>(sm, a.ctx, d, 0, w1, lw.out_b, X, d, u * 16);
    }
    ++phase; ct_arrive(sync, phase);
    if (b < nF1) ct_load_w(w2, lw.fc1_w, d, b * 32, 0);
    ct_wait(sync, phase, dead, a.err); CT_STAMP();
    // ---- ffn = gelu(fc1(LayerNorm(x)))
    for (int u = b; u < nF1; u += G) {
      if (u != b) { __syncthreads(); ct_load_w(w2, lw.fc1_w, d, u * 32, 0); }
      ct_unit_ln<MT, NKS, 2, EPI_BF16_GELU>(sm, X, d, lw.ln2_g, lw.ln2_b, a.eps, w2, lw.fc1_b, a.ffn, f, u * 32);
    }
    ++phase; ct_arrive(sync, phase);
    if (b < nF2) ct_load_w(w2, lw.fc2_w, f, (b % nPair) * 32, (b / nPair) * d);
    ct_wait(sync, phase, dead, a.err); CT_STAMP();
    // ---- x += fc2(ffn): four K-splits per feature pair; the last one to finish adds them in split order
    for (int u = b; u < nF2; u += G) {
      const int pair = u % nPair, split = u / nPair;
      if (u != b) { __syncthreads(); ct_load_w(w2, lw.fc2_w, f, pair * 32, split * d); }
      ct_unit_bf16<MT, NKS, 2, EPI_F32_PARTIAL>(sm, a.ffn, f, split * d, w2, nullptr, part + (size_t)split * part_stride, d, pair * 32);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                     // wave 0's partial stores are acknowledged
      if (tid == 0) {
        // relaxed on purpose (no cache maintenance): the partials are device-scope stores that have been acknowledged, the reader
        // uses device-scope loads
        const unsigned old = __hip_atomic_fetch_add(pair_cnt + pair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm.flag = old == 3u;
        if (old == 3u) CT_ST_RLX(pair_cnt + pair, 0u);
      }
      __syncthreads();
      if (sm.flag && tid < Mrows * 8) {                    // splitk_reduce_kernel's arithmetic on this pair's 32 features
        const int m = tid >> 3, c = tid & 7;
        const int o = (m * d + pair * 32 + c * 4) * 4;
        const rsrc_t rp = ct_rsrc(part), rx = ct_rsrc(X);
        float4 acc = CT_LDG(float4, lw.fc2_b + pair * 32 + c * 4);
        float4 pv[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) pv[sidx] = ct_ld_dev<float4>(rp, sidx * (int)part_stride * 4 + o);
        float4 r = ct_ld_dev<float4>(rx, o);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) { acc.x += pv[sidx].x; acc.y += pv[sidx].y; acc.z += pv[sidx].z; acc.w += pv[sidx].w; }
        r.x += acc.x; r.y += acc.y; r.z += acc.z; r.w += acc.w;
        ct_st_dev<float4>(rx, o, r);
      }
    }
    ++phase; ct_arrive(sync, phase);
    if (!last && b < nQ) ct_load_w(w1, a.layers[l + 1].qkv_w, d, b * 16, 0);
    ct_wait(sync, phase, dead, a.err); CT_STAMP();
  }
  // rearm for the next launch: every workgroup clears its own slot (the aggregator has read it for the last time: the last barrier
  // is open); the flags may still be polled by others, so they are cleared by whoever leaves last
  if (tid == 0) {
    CT_ST_RLX(sync + CT_SLOTS + b, 0u);
    const unsigned old = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)G - 1u) {
      for (int g = 0; g < 8; ++g) CT_ST_RLX(sync + CT_FLAGS + 16 * g, 0u);
      CT_ST_RLX(sync + 2, 0u);
    }
  }
}

// may the persistent trunk take this forward?  M = padded token rows (16 or 32)
bool chain_trunk_ok(int M, int d_model, int d_ffn, int n_heads) {
  static const int on = [] { const char* e = getenv("PGIBBS_CHAIN_TRUNK"); return e ? atoi(e) : 1; }();
  // d_model 1024 or 1280: the widths at which the multi-launch path splits fc2's K = 4 d_model four ways, like this kernel
  return on && (M == 16 || M == 32) && (d_model == 1024 || d_model == 1280) && d_ffn == 4 * d_model && n_heads * 64 == d_model;
}
size_t chain_trunk_sync_bytes() { return (size_t)(CT_SLOTS + CT_MAX_GRID) * 4; }   // exit counter, pair counters, flags, slots (zeroed once)
size_t chain_trunk_part_bytes(int M, int d_model) { return (size_t)4 * M * d_model * 4; }

// One workgroup per CU, all resident at once (the barriers spin).  Nothing in an ordinary launch guarantees that, so the grid is
// bounded per DEVICE by what the occupancy calculator says can be resident (CU count x workgroups per CU of this very kernel:
// 512 threads, ~65 KB of LDS, <= 128 VGPRs) and kChainTrunkUnfit sends the caller to the per-layer launches when not even one
// workgroup per CU fits.  What no query can see -- another process's persistent grid, a CU mask -- is caught by the barrier
// timeout (Engine::chain_check).
template <int MT, int NK>
static int chain_trunk_grid(int device) {
  static int cached[64];                                   // per device ordinal: 0 = not asked yet, < 0 = unfit
  if (device < 0 || device >= 64) return -1;
  if (cached[device] == 0) {
    hipDeviceProp_t p;
    int per_cu = 0;
    if (hipGetDeviceProperties(&p, device) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chain_trunk_kernel<MT, NK>, 512, 0) != hipSuccess || per_cu < 1)
      cached[device] = -1;
    else
      cached[device] = std::min(p.multiProcessorCount, CT_MAX_GRID);
  }
  return cached[device];
}

int launch_chain_trunk(hipStream_t s, const PgChainTrunkArgs& a, int M, int d_model) {
  static const int g_env = [] { const char* e = getenv("PGIBBS_CHAIN_TRUNK_GRID"); return e ? atoi(e) : 0; }();
  if (a.n_layers <= 0) return 0;
  if (a.B * a.T > M || a.T > 32 || a.T < 1) return fail(1, "chain_trunk: shape");
  int dev = 0;
  PG_HIP(hipGetDevice(&dev));
  int n_cu;
#define PG_CT_GRID(MTV, NK) n_cu = chain_trunk_grid<MTV, NK>(dev)
  if (M == 16) { if (d_model / 256 == 4) PG_CT_GRID(1, 4); else PG_CT_GRID(1, 5); }
  else { if (d_model / 256 == 4) PG_CT_GRID(2, 4); else PG_CT_GRID(2, 5); }
#undef PG_CT_GRID
  if (n_cu < 1) return kChainTrunkUnfit;
  const int G = g_env > 0 && g_env <= n_cu ? g_env : n_cu;
  dim3 grid(G), block(512);
  note_kernel("chain_trunk (all layers, persistent)", G);
#define PG_CT(MTV, NK) hipLaunchKernelGGL((chain_trunk_kernel<MTV, NK>), grid, block, 0, s, a)
#define PG_CT_NK(MTV)                          \
  switch (d_model / 256) {                     \
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
