// Shared helpers for the gfx950 kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace pg {

// ---- errors ---------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define PG_HIP(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return ::pg::fail(2, std::string(#expr) + ": " + hipGetErrorString(_e));                        \
  } while (0)

// ---- bf16 -----------------------------------------------------------------------------------
typedef uint16_t bf16_t;  // raw bits

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
// round-to-nearest-even, NaN preserved
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  union { uint32_t u; float f; } x;
  x.f = f;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40);
  uint32_t r = 0x7fffu + ((x.u >> 16) & 1u);
  return (bf16_t)((x.u + r) >> 16);
}
#if defined(__HIPCC__)
// device: the cast lowers to gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even)
typedef __bf16 pg_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  pg_bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16_dev(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
#endif

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Rows of every activation buffer are padded to this many rows so the GEMM tiles never need a
// bounds check (pad rows hold finite garbage and are never read back).
constexpr int kRowPad = 256;

}  // namespace pg
