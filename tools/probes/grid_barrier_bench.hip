// PROBE, not part of libpgibbs.so (round 4, VERDICT r03 item 3: config 1 latency).  What does a device-wide barrier cost on the
// 8-XCD MI355X when one workgroup sits on every CU, and what does one "phase" of a persistent small-batch transformer layer cost
// around it (weights prefetched before the barrier, the activation rows read after it)?  Barrier forms:
//   0  one counter: every workgroup adds 1 (agent-scope release) and polls it
//   1  the same, polling every ~0.5 us instead of continuously
//   2  two levels: 8 group counters (blockIdx % 8 = the XCD under round-robin dispatch), the last arriver of a group adds to a
//      top counter, the last of those publishes the epoch in 8 per-group flags; workgroups poll their group's flag
//   3  no read-modify-write at all: every workgroup STORES the epoch into its own slot (16 slots per 64-byte line); one dedicated
//      workgroup polls the 256 slots (one dwordx4 per lane) and publishes the epoch in the 8 flags
//   6  form 5 without the aggregator: wave 0 of every workgroup polls all 256 slots itself
//   4, 5  forms 0 and 3 with relaxed atomics only: no buffer_wbl2 / buffer_inv (what a kernel would use whose inter-phase data
//      moves with sc1 loads and stores)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/grid_barrier_bench.hip -o build/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// sync block (uint32 words): [0] flat counter, [1] error, [16 + 16 g] group counters, [160] top counter, [192 + 16 g] flags,
// [512 .. 512 + 256) slots
#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

template <int BAR>
__device__ inline void grid_sync(unsigned* s, unsigned epoch, int G, int agg) {
  __syncthreads();
  const int b = blockIdx.x, g = b & 7;
  if (BAR == 4) {                                  // flat counter, relaxed: no buffer_wbl2 / buffer_inv (data moves with sc1 accesses)
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();
      while (LD(s) < epoch * (unsigned)G) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
    }
  } else if (BAR == 6) {                           // slots, relaxed, no aggregator: wave 0 of EVERY workgroup watches all slots
    if (threadIdx.x == 0) ST(s + 512 + b, epoch);
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const long long t0 = wall_clock64();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int w = 4 * l + i;
          if (w < G) ok = ok && LD(s + 512 + w) >= epoch;
        }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
    }
  } else if (BAR == 5) {                           // slots + aggregator, relaxed
    if (threadIdx.x == 0) ST(s + 512 + b, epoch);
    if (b == agg && threadIdx.x < 64) {
      const int l = threadIdx.x;
      const long long t0 = wall_clock64();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int w = 4 * l + i;
          if (w < G) ok = ok && LD(s + 512 + w) >= epoch;
        }
        if (__all(ok)) break;
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
      if (l < 8) ST(s + 192 + 16 * l, epoch);
    }
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();
      while (LD(s + 192 + 16 * g) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
    }
  } else if (BAR <= 1) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const long long t0 = wall_clock64();
      while (LD(s) < epoch * (unsigned)G) {
        __builtin_amdgcn_s_sleep(BAR == 1 ? 16 : 1);
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  } else if (BAR == 2) {
    if (threadIdx.x == 0) {
      const unsigned members = (unsigned)((G - g + 7) / 8);
      const unsigned old = __hip_atomic_fetch_add(s + 16 + 16 * g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == epoch * members) {
        const unsigned ngroups = G < 8 ? G : 8;
        const unsigned t = __hip_atomic_fetch_add(s + 160, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1 == epoch * ngroups) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          for (int i = 0; i < 8; ++i) ST(s + 192 + 16 * i, epoch);
        }
      }
      const long long t0 = wall_clock64();
      while (LD(s + 192 + 16 * g) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  } else {
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      ST(s + 512 + b, epoch);
    }
    if (b == agg) {
      if (threadIdx.x < 64) {                      // the aggregator: lane l watches slots 4l .. 4l+3
        const int l = threadIdx.x;
        const long long t0 = wall_clock64();
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int w = 4 * l + i;
            if (w < G) ok = ok && LD(s + 512 + w) >= epoch;
          }
          if (__all(ok)) break;
          if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        if (l < 8) ST(s + 192 + 16 * l, epoch);
      }
    }
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();
      while (LD(s + 192 + 16 * g) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 5000000) { s[1] = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
  __syncthreads();
}

// work 0: barriers only.  work 1: + every workgroup writes 2 KB before the barrier and reads xkb KB of fp32 rows after it.
// work 2: + 40 KB of weights per workgroup fetched BEFORE the barrier (fresh addresses every phase).  The aggregator workgroup of
// barrier form 3 does no work (the real kernel has idle workgroups in every phase).
template <int BAR, int WORK>
__global__ __launch_bounds__(512) void phases_kernel(unsigned* sync, float* act, const uint4* w, unsigned w_mask, float* sink,
                                                     int n_phases, int xkb) {
  const int G = gridDim.x, tid = threadIdx.x;
  const int agg = G - 1;
  const bool worker = !((BAR == 3 || BAR == 5) && (int)blockIdx.x == agg);
  float accum = 0.f;
  unsigned wpos = blockIdx.x * 2560u;               // 40 KB = 2560 uint4 per workgroup and phase
  for (int p = 0; p < n_phases; ++p) {
    uint4 wreg[5];
    if (WORK >= 2 && worker) {
#pragma unroll
      for (int i = 0; i < 5; ++i) wreg[i] = w[(wpos + i * 512u + tid) & w_mask];
      wpos += (unsigned)G * 2560u;
    }
    if (WORK >= 1 && worker) {
      float* dst = act + (size_t)(p & 1) * 32 * 4096 + (size_t)(tid >> 4) * 4096 + (blockIdx.x % 256) * 16 + (tid & 15);
      *dst = accum + (float)p;
    }
    grid_sync<BAR>(sync, (unsigned)(p + 1), G, agg);
    if (WORK >= 1 && worker) {
      const float4* src = (const float4*)(act + (size_t)(p & 1) * 32 * 4096);
      const int n4 = xkb * 1024 / 16;
      float4 v[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) v[i] = (i * 512 + tid) < n4 ? src[i * 512 + tid] : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 18; ++i) accum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (WORK >= 2 && worker) {
#pragma unroll
      for (int i = 0; i < 5; ++i) accum += (float)(wreg[i].x ^ wreg[i].y ^ wreg[i].z ^ wreg[i].w);
    }
  }
  if (accum == 12345.678f) sink[blockIdx.x * 512 + tid] = accum;
}

// Form 7 (round 5): NO barrier -- the epoch travels inside the data (the "LL" protocol of the collective
// libraries).  A phase's output is xkb KB of payload as 8-byte units (4 bytes of payload + 4 bytes of epoch, one single-copy-atomic
// device-scope store each), spread over all workgroups; every workgroup then reads ALL units with device-scope 16-byte loads and
// re-reads until every unit shows this phase's epoch.  Two buffers alternate by phase parity (a workgroup can only start writing
// phase p + 2 after it has seen all of phase p + 1, which every other workgroup wrote after it had finished reading phase p).
// The 40 KB of weights per workgroup are fetched first, as in work mode 2.  What it measures: one store-to-load propagation + the
// polling traffic (G workgroups x 2 xkb KB per attempt) against the store-acknowledge + barrier + load of forms 4 / 5.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void ll_phases_kernel(unsigned* sync, v2u* act, const uint4* w, unsigned w_mask, float* sink, int n_phases,
                                                        int xkb) {
  const int G = gridDim.x, tid = threadIdx.x, b = blockIdx.x;
  const int NU = xkb * 256;                          // units = payload floats
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)act, 0, 0x7fffffff, 0x00020000);
  float accum = 0.f;
  unsigned wpos = blockIdx.x * 2560u;
  for (int p = 0; p < n_phases; ++p) {
    uint4 wreg[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) wreg[i] = w[(wpos + i * 512u + tid) & w_mask];
    wpos += (unsigned)G * 2560u;
    const unsigned epoch = (unsigned)(p + 1);
    const int base = (p & 1) * NU;
    for (int u = tid * G + b; u < NU; u += 512 * G) {
      const v2u val = {__float_as_uint(accum + (float)p), epoch};
      __builtin_amdgcn_raw_buffer_store_b64(val, rs, (base + u) * 8, 0, 16);          // sc1
    }
    // consume: thread t of every workgroup takes unit pairs t, t + 512, ...
    const int n2 = NU / 2;
    float got = 0.f;
    const long long t0 = wall_clock64();
    for (int q = tid; q < n2; q += 512 * 4) {
      v4u v[4];
      bool ok;
      do {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int qq = q + i * 512;
          v[i] = qq < n2 ? __builtin_amdgcn_raw_buffer_load_b128(rs, (base + 2 * qq) * 8, 0, 16) : (v4u){0u, epoch, 0u, epoch};
        }
        ok = true;
#pragma unroll
        for (int i = 0; i < 4; ++i) ok = ok && v[i][1] == epoch && v[i][3] == epoch;
        if (!ok && wall_clock64() - t0 > 5000000) { sync[1] = 1; ok = true; }
      } while (!ok);
#pragma unroll
      for (int i = 0; i < 4; ++i) got += __uint_as_float(v[i][0]) + __uint_as_float(v[i][2]);
    }
    accum += got * 1e-30f;
#pragma unroll
    for (int i = 0; i < 5; ++i) accum += (float)(wreg[i].x ^ wreg[i].y ^ wreg[i].z ^ wreg[i].w) * 1e-30f;
    __syncthreads();                                 // the workgroup's phase ends when all of its waves have their inputs
  }
  if (accum == 12345.678f) sink[blockIdx.x * 512 + tid] = accum;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int NCU = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, NCU);
  unsigned* sync; float *act, *sink; uint4* w;
  const size_t w_bytes = (size_t)1 << 30;
  CK(hipMalloc(&sync, 8192)); CK(hipMalloc(&act, 2 * 32 * 4096 * 4)); CK(hipMalloc(&sink, (size_t)NCU * 512 * 4)); CK(hipMalloc(&w, w_bytes));
  CK(hipMemset(act, 0, 2 * 32 * 4096 * 4)); CK(hipMemset(w, 1, w_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int NP = 1000;
  const unsigned w_mask = (unsigned)(w_bytes / 16 - 1);
  auto run = [&](int bar, int work, int xkb, int grid) {
    float best = 1e30f; unsigned herr[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(sync, 0, 8192));
      CK(hipEventRecord(e0));
#define L(B, W) if (bar == B && work == W) hipLaunchKernelGGL((phases_kernel<B, W>), dim3(grid), dim3(512), 0, 0, sync, act, w, w_mask, sink, NP, xkb)
      L(0, 0); L(0, 1); L(0, 2); L(1, 0); L(1, 2); L(2, 0); L(2, 1); L(2, 2); L(3, 0); L(3, 1); L(3, 2); L(4, 0); L(4, 2); L(5, 0); L(5, 2); L(6, 0); L(6, 2);
#undef L
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(herr, sync, 8, hipMemcpyDeviceToHost));
    }
    printf("barrier %d work %d grid %3d x-read %3d KB: %6.2f us per phase%s\n", bar, work, grid, xkb, best * 1e3f / NP, herr[1] ? "  [TIMEOUT FLAG SET]" : "");
    fflush(stdout);
  };
  for (int bar = 0; bar < 7; ++bar) { run(bar, 0, 0, NCU); run(bar, 0, 0, NCU / 2); run(bar, 0, 0, 64); }
  for (int bar : {0, 2, 3}) { run(bar, 1, 69, NCU); run(bar, 1, 138, NCU); run(bar, 2, 69, NCU); run(bar, 2, 138, NCU); }
  for (int bar : {4, 5, 6}) { run(bar, 2, 69, NCU); run(bar, 2, 138, NCU); }
  {
    v2u* act2; CK(hipMalloc(&act2, 2 * 138 * 256 * 8 * 2));
    for (int xkb : {8, 35, 69, 138}) for (int grid : {NCU, NCU / 2}) {
      float best = 1e30f; unsigned herr[2] = {0, 0};
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(sync, 0, 8192)); CK(hipMemset(act2, 0, 2 * 138 * 256 * 8 * 2));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(ll_phases_kernel, dim3(grid), dim3(512), 0, 0, sync, act2, w, w_mask, sink, NP, xkb);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        CK(hipMemcpy(herr, sync, 8, hipMemcpyDeviceToHost));
      }
      printf("epoch-in-data (no barrier) work 2 grid %3d payload %3d KB (%3d KB of units): %6.2f us per phase%s\n", grid, xkb, 2 * xkb, best * 1e3f / NP,
             herr[1] ? "  [TIMEOUT FLAG SET]" : "");
      fflush(stdout);
    }
  }
  return 0;
}
