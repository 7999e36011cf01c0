// Probe: where a wave of attention_kernel<18> spends its cycles at BASELINE config 2 (256 chains x T = 258, 20 heads).
// Builds the kernel with timestamp hooks (PG_ATT_PROF) and prints the mean shader-clock cycles per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPG_ATT_PROF -I include -I protein_gibbs_sampler_amd/csrc \
//       tools/probes/attention_phases.hip -o /tmp/attention_phases && /tmp/attention_phases
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "attention.hip"

namespace pg {
int fail(int code, const std::string& msg) { fprintf(stderr, "%s\n", msg.c_str()); return code; }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, T = 258, H = 20, d = H * 64;
  const size_t M = (size_t)B * T;
  pg::bf16_t *qkv, *ctx;
  (void)hipMalloc(&qkv, M * 3 * d * 2);
  (void)hipMalloc(&ctx, M * d * 2);
  std::vector<pg::bf16_t> h(M * 3 * d);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = pg::f32_to_bf16(((st >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.7f); }
  (void)hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const size_t n_wg = (size_t)B * H, n_stamp = n_wg * 4 * 8 * 8;
  unsigned long long* prof;
  (void)hipMalloc(&prof, n_stamp * 8);
  (void)hipMemset(prof, 0, n_stamp * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pg::pg_att_prof), &prof, sizeof(prof));
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int it = 0; it < 3; ++it) {
    (void)hipEventRecord(a);
    pg::launch_attention_bf16(nullptr, qkv, ctx, B, T, H, 3 * d, d, d, 2 * d, nullptr, -1);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("launch %d: %.3f ms (with timestamp stores)\n", it, ms);
  }
  std::vector<unsigned long long> p(n_stamp);
  (void)hipMemcpy(p.data(), prof, n_stamp * 8, hipMemcpyDeviceToHost);
  // slot 7: 0 = kernel entry, 1 = staging stores issued, 2 = after the barrier, 3 = wave done
  double stage = 0, barrier = 0, total = 0, ph[4] = {0, 0, 0, 0}, gap = 0;
  size_t nb = 0, nw = 0;
  for (size_t wg = 0; wg < n_wg; ++wg)
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* q = &p[((wg * 4 + w) * 8) * 8];
      const unsigned long long* e = q + 7 * 8;
      stage += (double)(e[1] - e[0]); barrier += (double)(e[2] - e[1]); total += (double)(e[3] - e[0]); ++nw;
      const int nblk = w == 0 ? 5 : 4;
      for (int s = 0; s < nblk; ++s) {
        const unsigned long long* t = q + s * 8;
        for (int i = 0; i < 4; ++i) ph[i] += (double)(t[i + 1] - t[i]);
        ++nb;
      }
    }
  printf("per wave: staging (loads + LDS writes) %.0f cycles, barrier wait %.0f, whole wave %.0f\n", stage / nw, barrier / nw, total / nw);
  printf("per 16-query block: S = K.Q^T %.0f | softmax %.0f | P.V %.0f | store + next %.0f  (sum %.0f)\n", ph[0] / nb, ph[1] / nb,
         ph[2] / nb, ph[3] / nb, (ph[0] + ph[1] + ph[2] + ph[3]) / nb);
  // workgroup residency: first entry to last exit
  double wg_cycles = 0;
  for (size_t wg = 0; wg < n_wg; ++wg) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned long long* e = &p[((wg * 4 + w) * 8 + 7) * 8];
      if (e[0] < t0) t0 = e[0];
      if (e[3] > t1) t1 = e[3];
    }
    wg_cycles += (double)(t1 - t0);
  }
  printf("workgroup residency %.0f cycles; %zu workgroups / (256 CUs x 2) x residency = %.3f ms at 2.4 GHz\n", wg_cycles / n_wg, n_wg,
         wg_cycles / n_wg * n_wg / 512 / 2.4e6);
  (void)gap;
  return 0;
}
