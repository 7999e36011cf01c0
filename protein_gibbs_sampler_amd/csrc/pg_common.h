// Shared helpers for the gfx950 kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace pg {

// ---- errors ---------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
// Which kernel did the dispatch pick?  Every GEMM launcher notes a short label ("pp256 165t", "tile64 x4 splits" ...) right before
// its launch; Engine::timed() attaches the labels noted inside a timed call to that call's record (pg_prof_get_kernels: the
// "kernel chosen per GEMM" column of tools/batch_sweep.py).  Thread-local, a few bytes per launch, no effect on the launch.
void note_kernel(const char* label, long tiles = -1, int splits = 0);
const std::string& noted_kernels();
void clear_noted_kernels();

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
// ---- operand flavour ------------------------------------------------------------------------------
// Every kernel family that stores or multiplies 16-bit operands (GEMMs, attention, LayerNorm / embedding outputs) is compiled
// TWICE from the same source: with bf16 operands (namespace pg::opbf16, inline: the default every unqualified call resolves to)
// and, with -DPG_F16, with IEEE fp16 operands (namespace pg::opf16; precision mode PG_PREC_F16).  v_mfma_f32_16x16x32_f16 runs
// at the bf16 rate and fp16 carries 3 more mantissa bits: the 16-bit roundings of weights, LayerNorm outputs, q/k/v, softmax
// numerators, context and FFN rows shrink 8x (profiles/r04_rounding_ablation.txt).  The flavour-dependent primitives are the
// four below; everything else moves 16-bit words without looking at them.  (The strict mode's split operands stay bf16 pairs.)
#ifdef PG_F16
#define PG_OPS_NS_OPEN namespace opf16 {
#else
#define PG_OPS_NS_OPEN inline namespace opbf16 {
#endif
#define PG_OPS_BEGIN namespace pg { inline namespace opbf16 {} namespace opf16 {} PG_OPS_NS_OPEN
#define PG_OPS_END } }
#if defined(__HIPCC__)
inline namespace opbf16 {}
namespace opf16 {}
PG_OPS_NS_OPEN
typedef __attribute__((ext_vector_type(8))) __bf16 pg_op16x8_t;      // 8 operand values = one 16-byte fragment (raw 16-bit lanes)
#ifdef PG_F16
// device: v_cvt_pk_f16_f32 (round-to-nearest-even).  No saturation: |x| > 65504 becomes inf -- the 16-bit tensors of this forward
// (LayerNorm outputs, q/k/v, softmax numerators in [0, 1], context, GELU(fc1)) stay orders of magnitude below that.
typedef _Float16 pg_f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 pg_f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {
  pg_f16x2_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_op16_dev(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float op16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float mfma_op16(pg_op16x8_t a, pg_op16x8_t b,
                                                                               __attribute__((ext_vector_type(4))) float c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pg_f16x8_t, a), __builtin_bit_cast(pg_f16x8_t, b), c, 0, 0, 0);
}
#else
// device: the cast lowers to gfx950's v_cvt_pk_bf16_f32 (round-to-nearest-even)
typedef __bf16 pg_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {
  pg_bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f32_to_op16_dev(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float op16_to_f32(bf16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ __attribute__((ext_vector_type(4))) float mfma_op16(pg_op16x8_t a, pg_op16x8_t b,
                                                                               __attribute__((ext_vector_type(4))) float c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif
}  // the flavour namespace

// Reductions over the four lanes {l, l^16, l^32, l^48} (the four 16-lane rows of a wave = the fq groups of an MFMA
// fragment) with gfx950's row-swap VALU instructions instead of two ds_bpermute round trips through the LDS pipe:
// v_permlane16_swap(a, b) swaps the odd rows of a with the even rows of b, v_permlane32_swap the upper half of a with the
// lower half of b; with a = b = x the two results hold (x[l], x[l ^ 16 / 32]) in every lane, in either order.
// Inline asm, not __builtin_amdgcn_permlane16/32_swap: hipcc 7.2 folds op(r[0], r[1]) of the builtin's two results to
// op(r[0], r[0]) (seen in the ISA; results off by the missing term).  The s_nop covers the "VALU write -> permlane read"
// hazard (2 wait states), which the compiler does not track through an asm statement.
__device__ __forceinline__ void rows_swap16(float x, float& a, float& b) {
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void rows_swap32(float x, float& a, float& b) {
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
#ifdef PG_ROWS4_SHFL   /* bisecting aid: the ds_bpermute form */
__device__ __forceinline__ float rows4_max(float x) { x = fmaxf(x, __shfl_xor(x, 16)); return fmaxf(x, __shfl_xor(x, 32)); }
__device__ __forceinline__ float rows4_sum(float x) { x += __shfl_xor(x, 16); return x + __shfl_xor(x, 32); }
#else
__device__ __forceinline__ float rows4_max(float x) {
  float a, b;
  rows_swap16(x, a, b);
  rows_swap32(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float rows4_sum(float x) {
  float a, b;
  rows_swap16(x, a, b);
  rows_swap32(a + b, a, b);
  return a + b;
}
#endif
#endif

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// Rows of every activation buffer are padded to this many rows so the GEMM tiles never need a
// bounds check (pad rows hold finite garbage and are never read back).
constexpr int kRowPad = 256;

}  // namespace pg
