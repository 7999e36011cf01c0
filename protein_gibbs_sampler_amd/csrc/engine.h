// Engine: weights of one masked protein LM resident on one MI355X + the Gibbs hot loop.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace pg {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  // grow-only; new storage is zero-filled (pad rows must hold finite values)
  int ensure(size_t need, hipStream_t s);
  void release();
  template <typename T> T* as() const { return (T*)p; }
};

struct Prof {
  struct Rec { int cls; hipEvent_t a, b; std::string kernels; };   // kernels: what the GEMM dispatch noted (pg::note_kernel)
  bool on = false;
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t get();
  void reset();
  void destroy();
};

// PC_GEMM_*: the four per-layer projections of a full-batch forward, timed apart (bench.py's per-kernel roofline); "gemm" = all five
enum ProfClass { PC_GEMM = 0, PC_ATTN, PC_LN, PC_EMBED, PC_HEAD, PC_SAMPLE, PC_GEMM_QKV, PC_GEMM_OUT, PC_GEMM_FC1, PC_GEMM_FC2, PC_COUNT };

struct DenseW {   // y = x W^T + b ; W bf16 [N][K], b fp32 [N].  Strict precision mode: w is [N][3K], each row the
  bf16_t* w = nullptr;   // K-concatenated split-bf16 operand [hi | lo | hi] (hi = bf16(W), lo = bf16(W - hi))
  float* b = nullptr;
  int N = 0, K = 0;
};
struct LnW { float* g = nullptr; float* b = nullptr; };

struct EsmLayer {
  LnW ln1, ln2;
  DenseW qkv, out, fc1, fc2;
  bf16_t* bias_kv16 = nullptr;   // ESM-1 (add_bias_kv): [bias_k | bias_v] as 16-bit operands of the engine's flavour ...
  float* bias_kv32 = nullptr;    // ... and fp32 (strict mode)
};
struct MsaLayer {
  LnW ln_row, ln_col, ln_ffn;
  DenseW row_qkv, row_out, col_qkv, col_out, fc1, fc2;
  DenseW col_qkv_hm;      // col_qkv with its rows grouped per head ([q_h | k_h | v_h]): operand of the fused column kernel (gemm_colattn.hip)
};

struct Engine {
  pg_model_config cfg;
  int device = 0;
  int precision = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<void*> owned;   // every hipMalloc'd weight block

  // weights
  float *embed = nullptr, *pos = nullptr, *msa_pos = nullptr, *embed_out = nullptr;
  float embed_scale() const { return cfg.arch == PG_ARCH_ESM1 ? sqrtf((float)cfg.d_model) : 1.0f; }
  LnW ln_before, ln_after, head_ln;
  DenseW head_dense;
  float* head_bias = nullptr;
  std::vector<EsmLayer> esm_layers;
  std::vector<MsaLayer> msa_layers;

  // workspace (grow-only)
  DevBuf x, h, qkv, ctx, ffn, sel_h, sel_g, logits, d_tokens, d_idx, d_samp_tok, d_samp_logits, d_rowmap, scratch;
  DevBuf x_sel, ctx_sel, h_sel, ffn_sel;   // last-layer pruning (compact rows)
  // x (+)= a W^T + b, then h = LayerNorm(x; ln): residual GEMM + LayerNorm kernel
  int resid_gemm_ln(const bf16_t* a, const DenseW& W, float* x, int M_rows, int64_t M_real, int lda, const LnW& ln, bf16_t* h,
                    float* ws = nullptr, size_t ws_bytes = 0, int prof_class = PC_GEMM, int colmajor_R = 0, int colmajor_C = 0);
  DevBuf tmp_idx, tmp_out;                 // batched generate_single on small MSAs: one template's step table / outputs
  DevBuf splitk;                           // fp32 partial maps of split-K fc2 GEMMs (small batches)
  bool esm_pad_in_batch = false;           // set by the host-token entry points: some token is <pad> -> key-padding mask
  float* splitk_ws(int rows, int n, int64_t batch_rows);
  int64_t job_items = 0;                   // pg_engine_set_job_items: batch items of the whole (multi-GPU) job, 0 = this call
  int64_t order_items = 0;                 // > 0: order-changing kernel choices as for a job of this many items (batched generate_single: 1)
  int64_t job_batch(int64_t B) const { return order_items > 0 ? order_items : (job_items > B ? job_items : B); }
  int64_t batch_rows = 0;                  // token rows of the JOB's forward (job_batch(B) x rows per item; set by the trunks)
  int sel_gemm_rows(int64_t n_sel, int64_t Np) const;
  // launch-bound (small) Gibbs loops: one iteration captured as a hipGraph and replayed; the iteration number lives in
  // d_iter on the device, so the same graph serves every iteration
  DevBuf d_iter;
  // persistent single-chain trunk (chain_trunk.hip): device table of the layers' weight pointers, barrier block, fc2 partials,
  // and a host-pinned error word the kernel sets when a device-wide barrier timed out
  PgChainLayerW* chain_layers = nullptr;
  DevBuf chain_sync, chain_part;
  unsigned* chain_err = nullptr;
  // after a stream synchronisation: PG_ERR_HIP if chain_err is set.  The kernel is then switched off for the rest of this engine's
  // life (the device is evidently shared with another persistent grid) and chain_retry tells the host-buffer entry points, whose
  // inputs are intact, to run the call again on the per-layer launches
  int chain_check();
  bool chain_disabled = false, chain_retry = false;
  // Device-pointer Gibbs calls (pg_esm_gibbs_run_device) are asynchronous and overwrite the caller's tokens in place, so a barrier
  // timeout is only seen after the damage.  Every such call that may take the persistent launch is therefore logged with a
  // snapshot of its (<= 32) token rows; when a later synchronisation finds the error word set, chain_check() switches the kernel
  // off, restores the first snapshot of every token buffer and runs the logged calls again, in order, on the per-layer launches.
  struct ChainCall {
    int32_t* d_tok; int B, T; const int32_t* d_idx; int n_iters, P; pg_sample_params sp; float* lg; int32_t* st; size_t snap_off;
  };
  std::vector<ChainCall> chain_log;
  bool chain_replay_ok = false;                              // set by chain_check(): the logged calls were run again successfully
  DevBuf chain_snap;
  static constexpr size_t kChainSnapBytes = 128, kChainLogMax = 64;
  bool chain_may_run(int B, int T) const;                    // would a forward of this shape take the persistent launch?
  int64_t batch_rows_for(int B, int T) const { return job_batch(B) * T; }
  int chain_log_call(int32_t* d_tok, int B, int T, const int32_t* d_idx, int n_iters, int P, const pg_sample_params* sp, float* lg,
                     int32_t* st);
  // host-pinned word the draw / log-probability kernels set when a logit row is not finite (an fp16 operand that overflowed
  // upstream, or broken weights).  After a stream synchronisation: PG_ERR_RANGE once, the word cleared.
  unsigned* range_err = nullptr;
  int range_check();
  int finish_check() { int rc = chain_check(); return rc ? rc : range_check(); }      // right after a stream synchronisation
  hipGraphExec_t graph_exec = nullptr;
  std::vector<uint8_t> graph_key;
  int64_t stat_graph_captures = 0, stat_graph_replays = 0;     // pg_engine_get_stat
  // strict precision mode (PG_PREC_FP32): h, ctx, ffn, sel_h hold [lo | hi | hi] rows (3x wide); fp32 fc1 output;
  // row-attention scores (also the bf16 mode's wide-alignment fallback)
  DevBuf ffn_f32, scores;
  bool strict() const { return precision == PG_PREC_FP32; }
  bool esm1() const { return cfg.arch == PG_ARCH_ESM1; }      // ESM-1 differences: see pgibbs.h PG_ARCH_ESM1
  // out[Mp][N] fp32 (=|+=) X.W^T + b with X = xh + xl, W = wh + wl as ONE bf16 GEMM over K' = 3K:
  // [xl | xh | xh] . [wh | wl | wh]^T = xl.wh + xh.wl + xh.wh  (the dropped xl.wl term is ~2^-17 relative)
  int dense3(const bf16_t* x3, const DenseW& W, float* out, int Mp, bool accumulate);
  int dense3_gelu(const bf16_t* x3, const DenseW& W, int Mp, const DenseW* next = nullptr);   // fc1: ffn (operand rows) = split3(gelu(.)); next = the projection that reads them
  bool dense3_wants_dup(const DenseW& W, int Mp, bool gelu = false) const;   // does this projection read the duplicate hi block of its operand rows?
  Prof prof;

  ~Engine();
  int init(const pg_model_config* c, const pg_tensor* tensors, int n_tensors, int device_ordinal, int precision);

  // ---- forward pieces (all on `stream`, device pointers) ----
  // tokens -> x (before ln_after).  With a selection (sel_idx != nullptr, bf16 mode) the LAST layer's out-proj, LN2 and
  // FFN run only on the selected rows and x_sel [n_sel][d] holds their residual stream (exact: nothing else reads
  // the last layer's output); head_compact() then consumes x_sel.
  int esm_trunk(const int32_t* d_tok, int B, int T, const int32_t* sel_idx = nullptr, int P = 0, int64_t n_sel = 0,
                const int32_t* d_iter = nullptr);
  int head(const int32_t* d_idx, const int32_t* d_row_map, int P, int width, int64_t n_sel, float* d_logits,
           const float* x_src = nullptr);
  int esm_gibbs_device(int32_t* d_tok, int B, int T, const int32_t* d_idx, int n_iters, int P, const pg_sample_params* sp,
                       float* d_samp_logits, int32_t* d_samp_tok);
  // tokens[B][R][C] -> x; with a selection the LAST layer's column out-projection and FFN run on the selected rows only
  // (x_sel), as in esm_trunk
  int msa_trunk(const int32_t* d_tok, int B, int R, int C, const int32_t* sel_idx = nullptr, const int32_t* sel_row_map = nullptr,
                int P = 0, int64_t n_sel = 0);
  int msa_gibbs_device(int32_t* d_tok, int B, int R, int C, const int32_t* d_idx, int n_iters, int P,
                       const pg_sample_params* sp, float* d_samp_logits, int32_t* d_samp_tok);
  // generate_single on B templates of equal shape: step s masks row mask_row of every template and samples row target_row of
  // template b at d_step_idx[s][b][P_max] with sp[b]
  int msa_single_device(int32_t* d_tok, int B, int R, int C, int mask_row, int target_row, const int32_t* d_step_idx,
                        const int32_t* step_sample_flag_host, int n_steps, int P_max, const pg_sample_params* sp,
                        float* d_samp_logits, int32_t* d_samp_tok);

  // timing helper
  template <typename F> int timed(int cls, F&& f);
};

// state-dict lookup used by init
struct TensorMap {
  std::map<std::string, const pg_tensor*> m;
  const float* get(const std::string& name, int64_t numel, std::string& err) const;
};

}  // namespace pg

struct pg_engine {
  pg::Engine e;
};
