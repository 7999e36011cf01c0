// Launchers of the gfx950 kernels (one translation unit per kernel family).
#pragma once
#include "pg_common.h"

#include "../../include/pgibbs.h"

namespace pg {

// a "sequence" for the attention kernels: sequence s covers token rows
//   (s / inner_count) * outer_rows + (s % inner_count) * inner_rows + t * row_step,  t = 0..T-1
struct SeqLayout { int inner_count, outer_rows, inner_rows, row_step; };

enum { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32_RESID = 2, EPI_F32 = 3, EPI_F32_GELU = 4,
       EPI_F32_PARTIAL = 5 /* internal: split-K partial sums, no bias */,
       EPI_SPLIT3_GELU = 6 /* strict mode fc1: erf-GELU, then the bf16 operand rows [lo | hi | hi] (ldo = 3 N) that fc2 reads; 16-wave kernel only */ };

// out[M][N] (+)= X[M][K] . W[N][K]^T + bias.  M a multiple of 16 up to 256 rows, of 128 beyond; N a multiple of 64; K of 64
// (activation buffers are padded to 256 rows: kernels may touch the padding rows of the last tile).
// ws (optional, fp32 scratch): lets a residual GEMM with few tiles and a deep K (fc2 of a small batch) run as parallel
// K-splits into ws followed by one fixed-order reduction into out -- bit-reproducible, no atomics.
int launch_gemm_bf16(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K,
                     int ldx, int ldw, int ldo, int epi, float* ws = nullptr, size_t ws_bytes = 0);
// variant: 1 = lockstep tiles only, 2 = default dispatch, 6 / 7 = force 64^2 / 128^2 tiles, 21.. = ablations of the
// ping-pong kernel (micro-benchmark entry)
int launch_gemm_bf16_variant(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N,
                             int K, int ldx, int ldw, int ldo, int epi, int variant, float* ws = nullptr, size_t ws_bytes = 0);

// weight-streaming GEMM with the preceding LayerNorm folded into its operand load (single chains; gemm_bf16.hip):
// out bf16 = LayerNorm(x fp32 [M][K]; gamma, beta) . W^T + bias (+ GELU); M = 16 or 32 rows, K = 256 .. 1280 in steps of 256
bool gemm_ln_skinny_ok(int M, int N, int K);
int launch_gemm_ln_skinny(hipStream_t s, const float* X, int ldx, const float* gamma, const float* beta, float eps, const bf16_t* W,
                          const float* bias, void* out, int M, int N, int K, int ldw, int ldo, int epi);
// the big-batch GEMM: whole rounds of 256 x 256 tiles + 64 x 64 tail tiles in one grid (gemm_bf16.hip); M, N multiples of 256
int launch_gemm_big(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldx,
                    int ldw, int ldo, int epi);
// its split of the rows: m-panels of 256 x 256 tiles (whole rounds of one tile per CU) + rows of 64 x 64 tail tiles
void gemm_big_geometry(int M, int N, int K, int* m_main_panels, int* tail_rows);
// the 16-wave 256x256 tile kernel (gemm_w16.hip): M (may be 0 with tail_rows > 0), N multiples of 256, K a multiple of 64;
// tail_rows rows of 64 x 64 tail tiles from row M on
int launch_gemm_w16(hipStream_t s, const bf16_t* X, const bf16_t* W, const float* bias, void* out, int M, int N, int K, int ldx,
                    int ldw, int ldo, int epi, int abl = 0, int tail_rows = 0);
// strict mode: the three split-bf16 products of a projection in one pass (gemm_w16.hip); X3 / W3 in the split operand layout,
// K = logical depth; bit-identical with launch_gemm_bf16 over K' = 3K on the same operands
int launch_gemm_split3(hipStream_t s, const bf16_t* X3, const bf16_t* W3, const float* bias, void* out, int M, int N, int K, int ldo,
                       int epi);      // dispatcher: fused kernel or the plain GEMM over K' = 3K (gemm_bf16.hip)
int launch_gemm_split3_w16(hipStream_t s, const bf16_t* X3, const bf16_t* W3, const float* bias, void* out, int M, int N, int K,
                           int ldo, int epi);

// fused attention, one (sequence, head) per workgroup; qkv rows are [q | k | v] with head h at h*64
// key_tok (optional): the int32 token buffer [n_seq][T]; keys whose token is pad_idx are masked (ragged batches)
int launch_attention_bf16(hipStream_t s, const bf16_t* qkv, bf16_t* ctx, int B, int T, int H, int ld_qkv, int ld_ctx,
                          int k_off, int v_off, const int32_t* key_tok = nullptr, int pad_idx = -1);
int launch_attention_seq_bf16(hipStream_t s, const bf16_t* qkv, bf16_t* ctx, int64_t n_seq, int T, int H, int ld_qkv,
                              int ld_ctx, int k_off, int v_off, SeqLayout sl, const int32_t* key_tok = nullptr,
                              int pad_idx = -1);
// MSA tied row attention: one C x C map per (msa, head) from scores summed over the R rows
// `partial` (optional fp32 scratch of partial_bytes) enables the split-R mode used when B*H is small
// order_bh (0 = B * H): the (msa, head) count the split-R decision is taken on (job-level, so that shards agree);
// msa_row_split_scratch_bytes: the fp32 scratch `partial` must offer for the split form (0 = this shape does not split)
size_t msa_row_split_scratch_bytes(int B, int R, int C, int H, int order_bh);
int launch_msa_row_attention_bf16(hipStream_t s, const bf16_t* qkv, bf16_t* ctx, int B, int R, int C, int H, int ld_qkv,
                                  int ld_ctx, int k_off, int v_off, float scale, float* partial = nullptr,
                                  size_t partial_bytes = 0, int order_bh = 0);

int launch_embed_ln(hipStream_t s, const int32_t* tokens, const float* embed, const float* pos, const float* msa_pos,
                    const float* gamma, const float* beta, float* x, int64_t n_tok, int T, int d, int pad_idx,
                    int mask_idx, int token_dropout, int rows_per_msa, float eps,
                    const float* gamma2 = nullptr, const float* beta2 = nullptr, bf16_t* h2 = nullptr);   // h2: also the first layer's LayerNorm of x
int launch_layernorm_bf16(hipStream_t s, const float* x, const float* gamma, const float* beta, bf16_t* h, int64_t M,
                          int d, float eps, bool split3 = false);   // split3: h rows are [lo | hi | hi], 3 d wide
int launch_layernorm_f32(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int64_t M, int d,
                         float eps);
int launch_gather_ln_bf16(hipStream_t s, const float* x, const int32_t* idx, const int32_t* row_map, int P, int width,
                          const float* gamma, const float* beta, bf16_t* h, int64_t n_sel, int d, float eps,
                          bool split3 = false);
// strict precision mode: fp32 [rows][K] -> K-concatenated split-bf16 operand bf16 [rows][3K], [lo | hi | hi] for an
// activation (optionally through erf-GELU), [hi | lo | hi] for a weight; fp32 GELU in place
int launch_split3_bf16(hipStream_t s, const float* src, bf16_t* dst, int64_t rows, int K, float scale, bool gelu, bool weight);
int launch_gelu_f32(hipStream_t s, float* p, int64_t n);
// strict precision mode attention: fp32 qkv in, softmax and accumulation in fp32 (VALU); ctx out as bf16 rows of ld_ctx
// values, or with split_d = d_model as the [lo | hi | hi] operand rows (ld_ctx = 3 d)
int launch_attention_f32(hipStream_t s, const float* qkv, bf16_t* ctx, int split_d, int64_t n_seq, int T, int H,
                         int ld_qkv, int ld_ctx, int k_off, int v_off, SeqLayout sl, const int32_t* key_tok = nullptr,
                         int pad_idx = -1);
// strict tied row attention; `scores` is an fp32 scratch of B*H*C rows of msa_row_scores_ld(C) floats
static inline int msa_row_scores_ld(int C) { return (C + 3) & ~3; }
int launch_msa_row_attention_f32(hipStream_t s, const float* qkv, float* scores, bf16_t* ctx, int split_d, int B, int R,
                                 int C, int H, int ld_qkv, int ld_ctx, int k_off, int v_off, float scale);
// d_iter (optional): device-side iteration counter; idx is then the base of a [n_iters][...] table (hipGraph replay)
int launch_gather_rows(hipStream_t s, const void* src, void* dst, const int32_t* idx, const int32_t* row_map, int P, int width,
                       int64_t n_sel, int row_bytes, const int32_t* d_iter = nullptr);
int launch_iter_counter(hipStream_t st, int32_t* d_iter, bool set, int value);
// graph state block (graph_state_bytes() bytes): d_iter[0] = iteration, then the call's sampling parameters -- the kernels of a
// captured iteration read both from the device, so ONE graph serves every iteration of every call of the same shape
size_t graph_state_bytes();
int launch_graph_state(hipStream_t st, int32_t* d_iter, int iteration, const pg_sample_params* p);
int launch_lm_tail(hipStream_t s, const float* g, const float* gamma, const float* beta, const float* embed,
                   const float* out_bias, float* logits, int64_t n, int d, int V, float eps);
int launch_f32_to_bf16(hipStream_t s, const float* src, bf16_t* dst, int64_t n, float scale);
int launch_bf16_to_f32(hipStream_t s, const bf16_t* src, float* dst, int64_t n);
int launch_scale_f32(hipStream_t s, float* p, int64_t n, float scale);

int launch_logprob_gather(hipStream_t st, const float* logits, int V, int compact, int width, const int32_t* idx,
                          const int32_t* row_map, const int32_t* targets, int64_t n_sel, int P, float* out);
int launch_mask_scatter(hipStream_t st, int32_t* tokens, int width, const int32_t* idx, const int32_t* row_map,
                        int64_t n_sel, int P, int mask_idx, const int32_t* d_iter = nullptr);
int launch_sample_writeback(hipStream_t st, int32_t* tokens, int width, const float* logits, int V, int compact,
                            const int32_t* idx, const int32_t* row_map, int64_t n_sel, int P, const pg_sample_params* p,
                            int iteration, int32_t* sampled_tokens, const int32_t* d_iter = nullptr);

}  // namespace pg
