// Probe: at which shader clock does the 16-wave GEMM main loop run?  Lane 0 of every workgroup stamps s_memtime (shader clock)
// and s_memrealtime (constant 100 MHz) at loop entry and exit; clock = d(memtime) / d(memrealtime) * 100 MHz.  Variants: the real
// kernel, MFMA only (ablation 1: no DMA in the loop), DMA + LDS reads only (ablation 2: no MFMA).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPG_W16_PROF -I include -I protein_gibbs_sampler_amd/csrc \
//       tools/probes/gemm_clock.hip -o /tmp/gemm_clock && /tmp/gemm_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <algorithm>

#include "gemm_w16.hip"

namespace pg {
int fail(int code, const std::string& msg) { fprintf(stderr, "%s\n", msg.c_str()); return code; }
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && argv[1][0] == 'z';      // all-zero operands: same instructions, (almost) no datapath toggling
  const int M = 66048, N = 3840, K = 1280;
  pg::bf16_t *x, *w, *out;
  float* bias;
  (void)hipMalloc(&x, (size_t)M * K * 2);
  (void)hipMalloc(&w, (size_t)N * K * 2);
  (void)hipMalloc(&out, (size_t)M * N * 2);
  (void)hipMalloc(&bias, N * 4);
  (void)hipMemset(bias, 0, N * 4);
  std::vector<pg::bf16_t> h((size_t)M * K);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = pg::f32_to_bf16(((st >> 8) * (1.0f / 8388608.0f) - 1.0f)); }
  if (zeros) std::fill(h.begin(), h.end(), (pg::bf16_t)0);
  (void)hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  printf("operands: %s\n", zeros ? "all zero" : "uniform random in [-1, 1)");
  const int n_tiles = (M / 256) * (N / 256);
  unsigned long long* prof;
  (void)hipMalloc(&prof, (size_t)n_tiles * 4 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pg::pg_w16_prof), &prof, sizeof(prof));
  std::vector<unsigned long long> p((size_t)n_tiles * 4);
  const int abls[3] = {0, 1, 2};
  const char* names[3] = {"real kernel", "no DMA in the loop (LDS reads + MFMA)", "no MFMA (DMA + LDS reads)"};
  for (int a = 0; a < 3; ++a) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) {                     // 20 back-to-back launches: the clock governor settles
      if (it == 19) (void)hipEventRecord(e0);
      pg::launch_gemm_w16(nullptr, x, w, bias, out, M, N, K, K, K, N, pg::EPI_BF16, abls[a]);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(p.data(), prof, p.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int t = 0; t < n_tiles; ++t) { cyc += (double)(p[t * 4 + 2] - p[t * 4]); wall += (double)(p[t * 4 + 3] - p[t * 4 + 1]); }
    printf("%-40s launch %.3f ms | main loop per tile: %.0f shader cycles in %.2f us -> %.0f MHz | %.0f cycles per K-step\n", names[a], ms,
           cyc / n_tiles, wall / n_tiles / 100.0, cyc / wall * 100.0, cyc / n_tiles / (K / 64));
  }
  return 0;
}
