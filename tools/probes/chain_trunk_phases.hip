// PROBE, not part of libpgibbs.so (round 4).  Where does the time of the persistent single-chain trunk go?  chain_trunk_kernel<2, 5>
// compiled with phase timestamps (PG_CT_TIMING), driven with synthetic weights of ESM-1b's layer shape; prints, per barrier of the
// layer in the middle of the run, the distribution over workgroups of: work time (barrier opened -> work done), publish time
// (release fence + atomic), wait time (arrival -> barrier opened), and when the slowest workgroup arrived.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPG_CT_TIMING -I include -I protein_gibbs_sampler_amd/csrc tools/probes/chain_trunk_phases.hip -o build/chain_trunk_phases
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <algorithm>
#include <vector>

#include "chain_trunk.hip"

namespace pg {
int fail(int code, const std::string& msg) { fprintf(stderr, "%s\n", msg.c_str()); return code; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int grid_arg = argc > 1 ? atoi(argv[1]) : 0;
  const int NL = 6, d = 1280, f = 5120, M = 32;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int G = grid_arg > 0 ? grid_arg : prop.multiProcessorCount;
  auto dev = [&](size_t bytes, int fill) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, fill, bytes)); return p; };
  std::vector<PgChainLayerW> tab(NL);
  for (int l = 0; l < NL; ++l) {
    PgChainLayerW& w = tab[l];
    // bf16 0x3c3c = 0.0115: small weights keep every value finite; gamma = beta = bias = 0 patterns are fine for timing
    w.ln1_g = (float*)dev(d * 4, 0); w.ln1_b = (float*)dev(d * 4, 0); w.ln2_g = (float*)dev(d * 4, 0); w.ln2_b = (float*)dev(d * 4, 0);
    w.qkv_w = (unsigned short*)dev((size_t)3 * d * d * 2, 0x3c); w.qkv_b = (float*)dev(3 * d * 4, 0);
    w.out_w = (unsigned short*)dev((size_t)d * d * 2, 0x3c); w.out_b = (float*)dev(d * 4, 0);
    w.fc1_w = (unsigned short*)dev((size_t)f * d * 2, 0x3c); w.fc1_b = (float*)dev(f * 4, 0);
    w.fc2_w = (unsigned short*)dev((size_t)d * f * 2, 0x3c); w.fc2_b = (float*)dev(d * 4, 0);
  }
  PgChainTrunkArgs a;
  a.layers = (PgChainLayerW*)dev(NL * sizeof(PgChainLayerW), 0);
  CK(hipMemcpy((void*)a.layers, tab.data(), NL * sizeof(PgChainLayerW), hipMemcpyHostToDevice));
  a.n_layers = NL; a.partial_last = 0; a.B = 1; a.T = 27; a.eps = 1e-5f;
  a.x = (float*)dev((size_t)M * d * 4, 0); a.qkv = (unsigned short*)dev((size_t)M * 3 * d * 2, 0); a.ctx = (unsigned short*)dev((size_t)M * d * 2, 0);
  a.ffn = (unsigned short*)dev((size_t)M * f * 2, 0); a.part = (float*)dev(pg::chain_trunk_part_bytes(M, d), 0); a.sync = (unsigned*)dev(pg::chain_trunk_sync_bytes(), 0);
  CK(hipHostMalloc((void**)&a.err, 4, hipHostMallocMapped)); *a.err = 0;
  long long* stamps = (long long*)dev((size_t)G * 1024 * 8, 0);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(pg::pg_ct_stamps), &stamps, sizeof(stamps)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((pg::chain_trunk_kernel<2, 5>), dim3(G), dim3(512), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("grid %d: %d layers in %.1f us = %.1f us per layer; timeout flag %u\n", G, NL, ms * 1e3, ms * 1e3 / NL, *a.err);
  std::vector<long long> h((size_t)G * 1024);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  // stamps per barrier: [work done, arrived, opened]; 5 barriers per layer; layer 3's five barriers
  const char* names[5] = {"LN1+QKV", "attention", "out-proj", "LN2+fc1", "fc2+reduce"};
  auto S = [&](int wg, int i) { return h[(size_t)wg * 1024 + i]; };
  for (int ph = 0; ph < 5; ++ph) {
    const int i0 = (3 * 5 + ph) * 3;                       // work done of this phase
    const int iprev = i0 - 1;                              // barrier before it opened
    std::vector<double> work, pub, wait;
    long long first_open = S(0, iprev), last_arr = 0, first_arr = 1LL << 62, last_open = 0;
    for (int w = 0; w < G; ++w) {
      work.push_back((S(w, i0) - S(w, iprev)) * 0.01);
      pub.push_back((S(w, i0 + 1) - S(w, i0)) * 0.01);
      wait.push_back((S(w, i0 + 2) - S(w, i0 + 1)) * 0.01);
      last_arr = std::max(last_arr, S(w, i0 + 1)); first_arr = std::min(first_arr, S(w, i0 + 1));
      last_open = std::max(last_open, S(w, i0 + 2)); first_open = std::min(first_open, S(w, iprev));
    }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("%-10s work min/med/max %5.2f %5.2f %5.2f | publish med/max %5.2f %5.2f | wait min/med/max %5.2f %5.2f %5.2f | "
           "first arrival +%.2f, last arrival +%.2f, last open +%.2f us after the previous barrier first opened\n",
           names[ph], q(work, 0), q(work, 0.5), q(work, 1), q(pub, 0.5), q(pub, 1), q(wait, 0), q(wait, 0.5), q(wait, 1),
           (first_arr - first_open) * 0.01, (last_arr - first_open) * 0.01, (last_open - first_open) * 0.01);
  }
  return 0;
}
