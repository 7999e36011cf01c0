// Probe: the LDS image written by one `buffer_load_dwordx4 ... lds` vs one `global_load_lds_dwordx4` (64 lanes x 16 B).
// hipcc --offload-arch=gfx950 -O2 lds_dma_image.hip -o lds_dma_image && ./lds_dma_image
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))
__global__ void probe(const unsigned* src, unsigned* out, int mode, int lds_base) {
  __shared__ __attribute__((aligned(16))) char smem[131072];
  for (int i = threadIdx.x; i < 131072 / 4; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  const int lane = threadIdx.x;
  if (mode == 0) {
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(smem + lds_base), 16, lane * 16, 0, 0, 0);
  } else {
    __builtin_amdgcn_global_load_lds(GLBP((const char*)src + lane * 16), LDSP(smem + lds_base), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = ((unsigned*)(smem + lds_base))[i - 128 < 0 ? i : i];
}
int main() {
  unsigned *src, *out;
  hipMalloc(&src, 4096); hipMalloc(&out, 4096);
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;             // dword i holds i: lane l's 16 B = dwords 4l..4l+3
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode)
    for (int base : {0, 16384, 65536, 98304 + 24576}) {
      probe<<<1, 64>>>(src, out, mode, base);
      hipMemcpy(h.data(), out, 2048, hipMemcpyDeviceToHost);
      int bad = 0, first = -1;
      for (int i = 0; i < 256; ++i) if (h[i] != (unsigned)i) { ++bad; if (first < 0) first = i; }
      printf("%s lds_base=%6d: %d of 256 dwords differ from the lane-linear image", mode ? "global_load_lds " : "buffer_load..lds", base, bad);
      if (bad) printf(" (first at dword %d: got %u; dwords 0..7 = %u %u %u %u %u %u %u %u)", first, h[first], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
      printf("\n");
    }
  return 0;
}
