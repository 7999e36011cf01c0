// PROBE, not part of libpgibbs.so (round 4, VERDICT r03 item 3: config 1 latency).  What does a device-wide barrier cost on the
// 8-XCD MI355X when one workgroup sits on every CU, and what does one "phase" of a persistent small-batch transformer layer cost
// around it (weights prefetched before the barrier, the activation rows read after it)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/grid_barrier_bench.hip -o build/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline void grid_sync(unsigned* ctr, unsigned target, unsigned* err) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 5000000) { *err = 1; break; }      // 50 ms of the 100 MHz clock: give up, never hang
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // one invalidate after the wait, not one per poll
  }
  __syncthreads();
}

// mode 0: barriers only.  mode 1: + every workgroup writes 2 KB before the barrier and reads XKB of fp32 rows after it (the rows
// other workgroups wrote).  mode 2: + 40 KB of weights per workgroup fetched BEFORE the barrier (fresh addresses every phase).
template <int MODE>
__global__ __launch_bounds__(512) void phases_kernel(unsigned* ctr, unsigned* err, float* act, const uint4* w, size_t w_words,
                                                     float* sink, int n_phases, int xkb) {
  const int G = gridDim.x, tid = threadIdx.x;
  float accum = 0.f;
  size_t wpos = (size_t)blockIdx.x * 2560;          // 40 KB = 2560 uint4 per workgroup and phase
  for (int p = 0; p < n_phases; ++p) {
    uint4 wreg[5];
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 5; ++i) wreg[i] = w[(wpos + (size_t)i * 512 + tid) % w_words];
      wpos += (size_t)G * 2560;
    }
    if (MODE >= 1) {                               // this phase's output tile: 32 rows x 16 features
      float* dst = act + (size_t)(p & 1) * 32 * 4096 + (size_t)(tid >> 4) * 4096 + (blockIdx.x % 256) * 16 + (tid & 15);
      *dst = accum + (float)p;
    }
    grid_sync(ctr, (unsigned)(p + 1) * G, err);
    if (MODE >= 1) {
      const float4* src = (const float4*)(act + (size_t)(p & 1) * 32 * 4096);
      const int n4 = xkb * 1024 / 16;
      float4 v[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) v[i] = (i * 512 + tid) < n4 ? src[i * 512 + tid] : make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 18; ++i) accum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 5; ++i) accum += (float)(wreg[i].x ^ wreg[i].y ^ wreg[i].z ^ wreg[i].w);
    }
  }
  if (accum == 12345.678f) sink[blockIdx.x * 512 + tid] = accum;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int G = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, G);
  unsigned* ctr; float *act, *sink; uint4* w;
  const size_t w_bytes = (size_t)1 << 30;
  CK(hipMalloc(&ctr, 8)); CK(hipMalloc(&act, 2 * 32 * 4096 * 4)); CK(hipMalloc(&sink, (size_t)G * 512 * 4)); CK(hipMalloc(&w, w_bytes));
  CK(hipMemset(act, 0, 2 * 32 * 4096 * 4)); CK(hipMemset(w, 1, w_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int NP = 1000;
  auto run = [&](int mode, int xkb, int grid) {
    float best = 1e30f; unsigned herr[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(ctr, 0, 8));
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(phases_kernel<0>, dim3(grid), dim3(512), 0, 0, ctr, ctr + 1, act, w, w_bytes / 16, sink, NP, xkb);
      if (mode == 1) hipLaunchKernelGGL(phases_kernel<1>, dim3(grid), dim3(512), 0, 0, ctr, ctr + 1, act, w, w_bytes / 16, sink, NP, xkb);
      if (mode == 2) hipLaunchKernelGGL(phases_kernel<2>, dim3(grid), dim3(512), 0, 0, ctr, ctr + 1, act, w, w_bytes / 16, sink, NP, xkb);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(herr, ctr, 8, hipMemcpyDeviceToHost));
    }
    printf("mode %d grid %3d x-read %3d KB: %.2f us per phase%s\n", mode, grid, xkb, best * 1e3f / NP, herr[1] ? "  [TIMEOUT FLAG SET]" : "");
  };
  run(0, 0, G); run(0, 0, G / 2); run(0, 0, 64);
  run(1, 16, G); run(1, 69, G); run(1, 138, G);
  run(2, 69, G); run(2, 138, G);
  return 0;
}
