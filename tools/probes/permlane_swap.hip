// Probe: lane images of v_permlane16_swap / v_permlane32_swap with both operands = lane id.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/permlane_swap.hip -o /tmp/permlane_swap && /tmp/permlane_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
  const unsigned l = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
  auto b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
  out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 1024);
  probe<<<1, 64>>>(d);
  unsigned h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap vdst'", "permlane16_swap src0'", "permlane32_swap vdst'", "permlane32_swap src0'"};
  for (int k = 0; k < 4; ++k) {
    printf("%s (vdst = lane, src0 = lane + 100):\n", names[k]);
    for (int r = 0; r < 4; ++r) { printf("  row %d:", r); for (int i = 0; i < 16; ++i) printf(" %3u", h[k * 64 + r * 16 + i]); printf("\n"); }
  }
  return 0;
}
