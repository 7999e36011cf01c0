// Launchers of the gfx950 kernels (one translation unit per kernel family).
#pragma once
#include "pg_common.h"

#include "../../include/pgibbs.h"

namespace pg {

// a "sequence" for the attention kernels: sequence s covers token rows
//   (s / inner_count) * outer_rows + (s % inner_count) * inner_rows + t * row_step,  t = 0..T-1
// (defined at global scope and aliased: argument-dependent lookup on a pg:: type would make calls inside pg::opf16 ambiguous
// with the inline default flavour)
}  // namespace pg
// launch_chain_trunk's answer when the persistent grid cannot be co-resident on the device (occupancy): take the per-layer launches
constexpr int kChainTrunkUnfit = -1;
struct PgSeqLayout { int inner_count, outer_rows, inner_rows, row_step; };
// the persistent single-chain trunk (chain_trunk.hip): one layer's weights as device pointers, and the launch's arguments
struct PgChainLayerW {
  const float *ln1_g, *ln1_b;
  const unsigned short* qkv_w; const float* qkv_b;     // [3 d][d] 16-bit operands (q rows pre-scaled), fp32 bias
  const unsigned short* out_w; const float* out_b;
  const float *ln2_g, *ln2_b;
  const unsigned short* fc1_w; const float* fc1_b;
  const unsigned short* fc2_w; const float* fc2_b;
};
struct PgChainTrunkArgs {
  const PgChainLayerW* layers;   // device array [n_layers]
  int n_layers;                  // layers to run, from layers[0]
  int partial_last;              // 1: the last of them stops after its attention (the pruned tail runs on the selected rows elsewhere)
  int B, T;                      // chains x tokens, B * T <= 32 token rows
  float* x;                      // [M][d] fp32 residual stream, in / out
  unsigned short* qkv;           // [M][3 d]
  unsigned short* ctx;           // [M][d]
  unsigned short* ffn;           // [M][4 d]
  float* part;                   // [4][M][d] fp32: fc2's K-split partial products
  unsigned* sync;                // chain_trunk_sync_bytes() of zeroed device memory
  unsigned* err;                 // host-visible word, set to 1 when a device-wide barrier timed out (results are then invalid)
  float eps;
};
namespace pg {
using SeqLayout = ::PgSeqLayout;

enum { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32_RESID = 2, EPI_F32 = 3, EPI_F32_GELU = 4,
       EPI_F32_PARTIAL = 5 /* internal: split-K partial sums, no bias */,
       EPI_SPLIT3_GELU = 6 /* strict mode fc1: erf-GELU, then the bf16 operand rows [lo | hi | hi] (ldo = 3 N) that fc2 reads; 16-wave kernel only */,
       EPI_SPLIT2_GELU = 7 /* the same rows without the duplicate hi block ([lo | hi | --], same ldo): for an fc2 that runs on the fused
                              three-product kernel, which never reads it (gemm_split3_fused) -- a third less epilogue traffic */ };

// ---- the operand-flavoured kernel families (pg_common.h): declared in both namespaces, defined once per flavour ----
inline namespace opbf16 {
#include "kernels_ops.inc"
}
namespace opf16 {
#include "kernels_ops.inc"
}

// ---- flavour-independent launchers (sample.hip) ----
int launch_iter_counter(hipStream_t st, int32_t* d_iter, bool set, int value);
// graph state block (graph_state_bytes() bytes): d_iter[0] = iteration, then the call's sampling parameters -- the kernels of a
// captured iteration read both from the device, so ONE graph serves every iteration of every call of the same shape
size_t graph_state_bytes();
int launch_graph_state(hipStream_t st, int32_t* d_iter, int iteration, const pg_sample_params* p);
int launch_logprob_gather(hipStream_t st, const float* logits, int V, int compact, int width, const int32_t* idx,
                          const int32_t* row_map, const int32_t* targets, int64_t n_sel, int P, float* out,
                          unsigned* nonfinite = nullptr);     // nonfinite: device-visible word set to 1 when a logit row holds NaN / inf
int launch_mask_scatter(hipStream_t st, int32_t* tokens, int width, const int32_t* idx, const int32_t* row_map,
                        int64_t n_sel, int P, int mask_idx, const int32_t* d_iter = nullptr);
int launch_sample_writeback(hipStream_t st, int32_t* tokens, int width, const float* logits, int V, int compact,
                            const int32_t* idx, const int32_t* row_map, int64_t n_sel, int P, const pg_sample_params* p,
                            int iteration, int32_t* sampled_tokens, const int32_t* d_iter = nullptr, unsigned* nonfinite = nullptr);

}  // namespace pg
