// Engine implementation: weight upload (fp32 state dict -> device bf16/fp32), the ESM-1b forward as a
// chain of gfx950 kernels on one HIP stream, and the device-resident Gibbs loop
//   mask scatter -> forward -> LM head at the sampled rows -> draw + write-back
// which replaces the reference's per-iteration Python loop
// (/root/reference/src/pgen/esm_sampler.py:209-234) with zero host round trips per iteration.
#include "engine.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gemm_epilogue.h"

namespace pg {

// the operand-flavoured launchers (pg_common.h): fp16 operands in PG_PREC_F16, bf16 otherwise (the strict mode's split operands
// are bf16 pairs)
#define OPS(fn, ...) (precision == PG_PREC_F16 ? opf16::fn(__VA_ARGS__) : opbf16::fn(__VA_ARGS__))

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const char* last_error_cstr() { return g_last_error.c_str(); }

static thread_local std::string g_noted;
void note_kernel(const char* label, long tiles, int splits) {
  if (g_noted.size() > 512) return;
  if (!g_noted.empty()) g_noted += " + ";
  g_noted += label;
  if (tiles >= 0) g_noted += " " + std::to_string(tiles) + "t";
  if (splits > 1) g_noted += " x" + std::to_string(splits) + "k";
}
const std::string& noted_kernels() { return g_noted; }
void clear_noted_kernels() { g_noted.clear(); }

// ------------------------------------------------------------------------------------------------
// bumped whenever any workspace buffer is (re)allocated: a captured graph holds raw pointers and must be re-captured
static uint64_t g_alloc_epoch = 0;

int DevBuf::ensure(size_t need, hipStream_t s) {
  if (need <= bytes) return 0;
  ++g_alloc_epoch;
  size_t cap = need + need / 8 + 4096;
  void* np = nullptr;
  PG_HIP(hipMalloc(&np, cap));
  PG_HIP(hipMemsetAsync(np, 0, cap, s));
  if (p) {
    PG_HIP(hipStreamSynchronize(s));
    PG_HIP(hipFree(p));
  }
  p = np;
  bytes = cap;
  return 0;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  bytes = 0;
}

hipEvent_t Prof::get() {
  if (used == pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    pool.push_back(e);
  }
  return pool[used++];
}
void Prof::reset() {
  recs.clear();
  used = 0;
}
void Prof::destroy() {
  for (auto e : pool) (void)hipEventDestroy(e);
  pool.clear();
  reset();
}

template <typename F> int Engine::timed(int cls, F&& f) {
  if (!prof.on) return f();
  hipEvent_t a = prof.get(), b = prof.get();
  clear_noted_kernels();
  PG_HIP(hipEventRecord(a, stream));
  int rc = f();
  PG_HIP(hipEventRecord(b, stream));
  prof.recs.push_back({cls, a, b, noted_kernels()});
  return rc;
}

const float* TensorMap::get(const std::string& name, int64_t numel, std::string& err) const {
  auto it = m.find(name);
  if (it == m.end()) {
    err = "missing tensor '" + name + "'";
    return nullptr;
  }
  if (it->second->numel != numel) {
    err = "tensor '" + name + "' has " + std::to_string(it->second->numel) + " elements, expected " + std::to_string(numel);
    return nullptr;
  }
  return it->second->data;
}

Engine::~Engine() {
  if (device >= 0) (void)hipSetDevice(device);
  for (void* p : owned) (void)hipFree(p);
  DevBuf* bufs[] = {&x, &h, &qkv, &ctx, &ffn, &sel_h, &sel_g, &logits, &d_tokens, &d_idx, &d_samp_tok, &d_samp_logits,
                    &d_rowmap, &scratch, &d_iter, &tmp_idx, &tmp_out, &x_sel, &ctx_sel, &h_sel, &ffn_sel, &ffn_f32, &scores, &splitk,
                    &chain_sync, &chain_part, &chain_snap};
  for (DevBuf* b : bufs) b->release();
  if (chain_err) (void)hipHostFree(chain_err);
  if (range_err) (void)hipHostFree(range_err);
  prof.destroy();
  if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
  if (own_stream) (void)hipStreamDestroy(own_stream);
}

// ------------------------------------------------------------------------------------------------
// weight upload
// ------------------------------------------------------------------------------------------------
namespace {
struct Uploader {
  Engine* e;
  const TensorMap* tm;
  std::string err;

  float* f32(const std::string& name, int64_t numel) {
    const float* src = tm->get(name, numel, err);
    if (!src) return nullptr;
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)numel * 4) != hipSuccess) { err = "hipMalloc failed for " + name; return nullptr; }
    e->owned.push_back(d);
    if (hipMemcpy(d, src, (size_t)numel * 4, hipMemcpyHostToDevice) != hipSuccess) { err = "H2D failed for " + name; return nullptr; }
    return (float*)d;
  }
  // concatenate several [n_i][K] fp32 matrices (each scaled) into one bf16 [sum n_i][K] device matrix + fp32 bias
  bool dense(DenseW& out, const std::vector<std::string>& prefixes, const std::vector<float>& scales, int n_each, int K) {
    const int parts = (int)prefixes.size();
    const int64_t N = (int64_t)n_each * parts;
    void *dw = nullptr, *db = nullptr, *tmp = nullptr;
    const int Kw = e->strict() ? 3 * K : K;      // strict: rows are [hi | lo | hi]
    if (hipMalloc(&dw, (size_t)N * Kw * 2) != hipSuccess || hipMalloc(&db, (size_t)N * 4) != hipSuccess ||
        hipMalloc(&tmp, (size_t)n_each * K * 4) != hipSuccess) { err = "hipMalloc failed for " + prefixes[0]; return false; }
    e->owned.push_back(dw);
    e->owned.push_back(db);
    for (int p = 0; p < parts; ++p) {
      const float* w = tm->get(prefixes[p] + ".weight", (int64_t)n_each * K, err);
      const float* b = w ? tm->get(prefixes[p] + ".bias", n_each, err) : nullptr;
      if (!w || !b) { (void)hipFree(tmp); return false; }
      bool ok = hipMemcpy(tmp, w, (size_t)n_each * K * 4, hipMemcpyHostToDevice) == hipSuccess;
      if (e->strict())
        ok = ok && launch_split3_bf16(e->stream, (const float*)tmp, (bf16_t*)dw + (size_t)p * n_each * Kw, n_each, K, scales[p],
                                      false, true) == 0;
      else
        ok = ok && (e->precision == PG_PREC_F16 ? opf16::launch_f32_to_bf16(e->stream, (const float*)tmp, (bf16_t*)dw + (size_t)p * n_each * K, (int64_t)n_each * K, scales[p])
                                              : opbf16::launch_f32_to_bf16(e->stream, (const float*)tmp, (bf16_t*)dw + (size_t)p * n_each * K, (int64_t)n_each * K, scales[p])) == 0;
      ok = ok && hipStreamSynchronize(e->stream) == hipSuccess;
      ok = ok && hipMemcpy((float*)db + (size_t)p * n_each, b, (size_t)n_each * 4, hipMemcpyHostToDevice) == hipSuccess;
      if (ok && scales[p] != 1.0f) ok = launch_scale_f32(e->stream, (float*)db + (size_t)p * n_each, n_each, scales[p]) == 0 &&
                                        hipStreamSynchronize(e->stream) == hipSuccess;
      if (!ok) { err = "upload failed for " + prefixes[p]; (void)hipFree(tmp); return false; }
    }
    (void)hipFree(tmp);
    out.w = (bf16_t*)dw;
    out.b = (float*)db;
    out.N = (int)N;
    out.K = K;
    return true;
  }
  // ESM-1's add_bias_kv: [bias_k | bias_v] as fp32 (strict mode) and as 16-bit operands of the engine's flavour
  bool bias_kv(EsmLayer& L, const std::string& kname, const std::string& vname, int d) {
    const float* bk = tm->get(kname, d, err);
    const float* bv = bk ? tm->get(vname, d, err) : nullptr;
    if (!bk || !bv) return false;
    void *d32 = nullptr, *d16 = nullptr;
    if (hipMalloc(&d32, (size_t)2 * d * 4) != hipSuccess || hipMalloc(&d16, (size_t)2 * d * 2) != hipSuccess) { err = "hipMalloc failed for " + kname; return false; }
    e->owned.push_back(d32);
    e->owned.push_back(d16);
    bool ok = hipMemcpy(d32, bk, (size_t)d * 4, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy((float*)d32 + d, bv, (size_t)d * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && (e->precision == PG_PREC_F16 ? opf16::launch_f32_to_bf16(e->stream, (const float*)d32, (bf16_t*)d16, 2 * d, 1.f)
                                           : opbf16::launch_f32_to_bf16(e->stream, (const float*)d32, (bf16_t*)d16, 2 * d, 1.f)) == 0;
    ok = ok && hipStreamSynchronize(e->stream) == hipSuccess;
    if (!ok) { err = "upload failed for " + kname; return false; }
    L.bias_kv32 = (float*)d32;
    L.bias_kv16 = (bf16_t*)d16;
    return true;
  }
  // a [q | k | v] projection with its rows regrouped per head (the fused column QKV + attention kernel's weight operand)
  bool headmajor(DenseW& out, const DenseW& in, int H) {
    void *dw = nullptr, *db = nullptr;
    if (hipMalloc(&dw, (size_t)in.N * in.K * 2) != hipSuccess || hipMalloc(&db, (size_t)in.N * 4) != hipSuccess) { err = "hipMalloc failed (head-major projection)"; return false; }
    e->owned.push_back(dw);
    e->owned.push_back(db);
    const bool ok = (e->precision == PG_PREC_F16 ? opf16::launch_headmajor_qkv(e->stream, in.w, in.b, (bf16_t*)dw, (float*)db, H, in.K)
                                                 : opbf16::launch_headmajor_qkv(e->stream, in.w, in.b, (bf16_t*)dw, (float*)db, H, in.K)) == 0 &&
                    hipStreamSynchronize(e->stream) == hipSuccess;
    if (!ok) { err = "head-major projection copy failed"; return false; }
    out.w = (bf16_t*)dw;
    out.b = (float*)db;
    out.N = in.N;
    out.K = in.K;
    return true;
  }
  bool ln(LnW& out, const std::string& prefix, int d) {
    out.g = f32(prefix + ".weight", d);
    out.b = out.g ? f32(prefix + ".bias", d) : nullptr;
    return out.g && out.b;
  }
};
}  // namespace

int Engine::init(const pg_model_config* c, const pg_tensor* tensors, int n_tensors, int device_ordinal, int prec) {
  cfg = *c;
  precision = prec;
  device = -1;
  if (cfg.arch != PG_ARCH_ESM1B && cfg.arch != PG_ARCH_MSA1B && cfg.arch != PG_ARCH_ESM1) return fail(PG_ERR_INVALID, "unknown arch");
  if (prec != PG_PREC_BF16 && prec != PG_PREC_FP32 && prec != PG_PREC_F16) return fail(PG_ERR_INVALID, "unknown precision mode");
  if (cfg.d_model % 128 || cfg.d_ffn % 128 || cfg.n_heads * 64 != cfg.d_model)
    return fail(PG_ERR_INVALID, "d_model and d_ffn must be multiples of 128 and head dim must be 64");
  if (cfg.vocab < 1 || cfg.vocab > 64) return fail(PG_ERR_INVALID, "vocab must be in 1..64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(PG_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
  if (device_ordinal < 0 || device_ordinal >= ndev) return fail(PG_ERR_INVALID, "Invalid cuda device number: cuda:" + std::to_string(device_ordinal));
  PG_HIP(hipSetDevice(device_ordinal));
  device = device_ordinal;
  PG_HIP(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
  stream = own_stream;

  TensorMap tm;
  for (int i = 0; i < n_tensors; ++i)
    if (tensors[i].name) tm.m[tensors[i].name] = &tensors[i];
  Uploader up{this, &tm, ""};
  const int d = cfg.d_model, f = cfg.d_ffn, V = cfg.vocab;
  const float qs = 0.125f;  // head_dim^-0.5 = 64^-0.5, folded into W_q and b_q (exact in bf16)
  bool ok = true;
  ok = ok && (embed = up.f32("embed_tokens.weight", (int64_t)V * d));
  ok = ok && (pos = up.f32("embed_positions.weight", (int64_t)(cfg.max_positions + cfg.pad_idx + 1) * d));
  if (esm1()) {
    // untied output projection straight from the residual stream: logits = x embed_out^T + embed_out_bias
    ok = ok && (embed_out = up.f32("embed_out.weight", (int64_t)V * d));
    ok = ok && (head_bias = up.f32("embed_out.bias", V));
  } else {
    ok = ok && up.ln(ln_before, "emb_layer_norm_before", d);
    ok = ok && up.ln(ln_after, "emb_layer_norm_after", d);
    ok = ok && up.dense(head_dense, {"lm_head.dense"}, {1.0f}, d, d);
    ok = ok && up.ln(head_ln, "lm_head.layer_norm", d);
    ok = ok && (head_bias = up.f32("lm_head.bias", V));
  }
  if (ok && cfg.arch != PG_ARCH_MSA1B) {
    esm_layers.resize(cfg.n_layers);
    for (int i = 0; ok && i < cfg.n_layers; ++i) {
      const std::string p = "layers." + std::to_string(i) + ".";
      EsmLayer& L = esm_layers[i];
      ok = ok && up.ln(L.ln1, p + "self_attn_layer_norm", d);
      ok = ok && up.dense(L.qkv, {p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"}, {qs, 1.f, 1.f}, d, d);
      ok = ok && up.dense(L.out, {p + "self_attn.out_proj"}, {1.f}, d, d);
      ok = ok && up.ln(L.ln2, p + "final_layer_norm", d);
      ok = ok && up.dense(L.fc1, {p + "fc1"}, {1.f}, f, d);
      ok = ok && up.dense(L.fc2, {p + "fc2"}, {1.f}, d, f);
      if (ok && esm1()) ok = up.bias_kv(L, p + "self_attn.bias_k", p + "self_attn.bias_v", d);
    }
  } else if (ok) {
    ok = ok && (msa_pos = up.f32("msa_position_embedding", (int64_t)cfg.max_msa_rows * d));
    msa_layers.resize(cfg.n_layers);
    for (int i = 0; ok && i < cfg.n_layers; ++i) {
      const std::string p = "layers." + std::to_string(i) + ".";
      MsaLayer& L = msa_layers[i];
      const std::string r = p + "row_self_attention.", cpre = p + "column_self_attention.", ff = p + "feed_forward_layer.";
      ok = ok && up.ln(L.ln_row, r + "layer_norm", d);
      ok = ok && up.dense(L.row_qkv, {r + "layer.q_proj", r + "layer.k_proj", r + "layer.v_proj"}, {1.f, 1.f, 1.f}, d, d);
      ok = ok && up.dense(L.row_out, {r + "layer.out_proj"}, {1.f}, d, d);
      ok = ok && up.ln(L.ln_col, cpre + "layer_norm", d);
      ok = ok && up.dense(L.col_qkv, {cpre + "layer.q_proj", cpre + "layer.k_proj", cpre + "layer.v_proj"}, {qs, 1.f, 1.f}, d, d);
      if (ok && !strict()) ok = up.headmajor(L.col_qkv_hm, L.col_qkv, cfg.n_heads);
      ok = ok && up.dense(L.col_out, {cpre + "layer.out_proj"}, {1.f}, d, d);
      ok = ok && up.ln(L.ln_ffn, ff + "layer_norm", d);
      ok = ok && up.dense(L.fc1, {ff + "layer.fc1"}, {1.f}, f, d);
      ok = ok && up.dense(L.fc2, {ff + "layer.fc2"}, {1.f}, d, f);
    }
  }
  if (!ok) return fail(PG_ERR_WEIGHTS, up.err.empty() ? std::string("weight upload failed") : up.err);
  if (cfg.arch == PG_ARCH_ESM1B && !strict() && OPS(chain_trunk_ok, 32, d, f, cfg.n_heads)) {
    // the persistent single-chain trunk reads its weights through one device table
    std::vector<PgChainLayerW> tab(cfg.n_layers);
    for (int i = 0; i < cfg.n_layers; ++i) {
      const EsmLayer& L = esm_layers[i];
      tab[i] = {L.ln1.g, L.ln1.b, L.qkv.w, L.qkv.b, L.out.w, L.out.b, L.ln2.g, L.ln2.b, L.fc1.w, L.fc1.b, L.fc2.w, L.fc2.b};
    }
    void* dt = nullptr;
    PG_HIP(hipMalloc(&dt, tab.size() * sizeof(PgChainLayerW)));
    owned.push_back(dt);
    PG_HIP(hipMemcpy(dt, tab.data(), tab.size() * sizeof(PgChainLayerW), hipMemcpyHostToDevice));
    chain_layers = (PgChainLayerW*)dt;
    PG_HIP(hipHostMalloc((void**)&chain_err, sizeof(unsigned), hipHostMallocMapped));
    *chain_err = 0;
  }
  PG_HIP(hipHostMalloc((void**)&range_err, sizeof(unsigned), hipHostMallocMapped));
  *range_err = 0;
  PG_HIP(hipStreamSynchronize(stream));
  return PG_OK;
}

int Engine::range_check() {
  if (!range_err || !*range_err) return PG_OK;
  *range_err = 0;
  return fail(PG_ERR_RANGE, precision == PG_PREC_F16
                                ? "non-finite logits: a 16-bit tensor of the forward left the fp16 range (+-65504) -- run this model with "
                                  "precision bf16 (or fp32); the results of this call are invalid"
                                : "non-finite logits (NaN / inf in the weights or an overflow in the forward); the results of this call are invalid");
}

// esm_trunk's own predicate for the persistent launch (a call that would not take it is not worth a snapshot, a log entry and a
// forced synchronisation every kChainLogMax calls): no <pad> batch, the rows fit one or two 16-row MFMA tiles, and the
// LayerNorm-folding weight-streaming GEMMs accept the shape
bool Engine::chain_may_run(int B, int T) const {
  if (!(chain_layers && chain_err && !chain_disabled && !esm_pad_in_batch && T >= 1 && T <= 32 && (int64_t)B * T <= 32)) return false;
  const int64_t M = (int64_t)B * T;
  const int Mi = round_up((int)M, 16);
  const int d = cfg.d_model, f = cfg.d_ffn;
  if (batch_rows_for(B, T) > 2048 && Mi < 64) return false;                  // a tiny shard of a big job runs 64-row tiles (sel_gemm_rows)
  return gemm_ln_skinny_ok(Mi, 3 * d, d) && gemm_ln_skinny_ok(Mi, f, d) && OPS(chain_trunk_ok, Mi, d, f, cfg.n_heads);
}

int Engine::chain_log_call(int32_t* d_tok, int B, int T, const int32_t* d_idx_, int n_iters, int P, const pg_sample_params* sp,
                           float* lg, int32_t* st) {
  int rc;
  if (chain_log.size() >= kChainLogMax) {              // bound the log: check (and, if need be, recover) now
    PG_HIP(hipStreamSynchronize(stream));
    if ((rc = chain_check()) && !chain_replay_ok) return rc;
    chain_retry = false;
    if (chain_disabled) return PG_OK;
  }
  if ((rc = chain_snap.ensure(kChainSnapBytes * kChainLogMax, stream))) return rc;
  const size_t off = chain_log.size() * kChainSnapBytes;
  PG_HIP(hipMemcpyAsync((char*)chain_snap.p + off, d_tok, (size_t)B * T * 4, hipMemcpyDeviceToDevice, stream));
  chain_log.push_back({d_tok, B, T, d_idx_, n_iters, P, *sp, lg, st, off});
  return PG_OK;
}

// Call right after a synchronisation of `stream`.  PG_OK when no barrier of the persistent trunk timed out since the last check.
// Otherwise the kernel is switched off for this engine, the logged device-pointer calls are replayed on the per-layer launches
// (PG_OK if that was all there was to repair: pg_engine_synchronize), and a host-buffer entry point that is in flight is told to
// run again (chain_retry + PG_ERR_HIP: its inputs are intact in the caller's buffers).
int Engine::chain_check() {
  chain_replay_ok = false;
  if (!chain_err || !*chain_err) { chain_log.clear(); return PG_OK; }
  *chain_err = 0;
  // the timed-out launch may have left garbage activations behind, and the draw / log-probability kernels that ran on them may
  // have raised the non-finite flag: that verdict belongs to the abandoned launch, not to the replay (or the caller's retry) that
  // follows -- they raise it again if the logits really are non-finite
  if (range_err) *range_err = 0;
  if (chain_sync.p) (void)hipMemsetAsync(chain_sync.p, 0, chain_sync.bytes, stream);
  if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; graph_key.clear(); }   // it holds the persistent launch
  if (!chain_disabled)
    fprintf(stderr, "pgibbs: a device-wide barrier of the persistent single-chain trunk timed out (another persistent grid on this GPU?); "
                    "this engine uses the per-layer launches from now on\n");
  chain_disabled = true;
  chain_retry = true;
  if (!chain_log.empty()) {
    std::vector<ChainCall> log;
    log.swap(chain_log);
    std::vector<int32_t*> seen;
    for (const ChainCall& c : log) {
      if (std::find(seen.begin(), seen.end(), c.d_tok) == seen.end()) {
        seen.push_back(c.d_tok);
        PG_HIP(hipMemcpyAsync(c.d_tok, (char*)chain_snap.p + c.snap_off, (size_t)c.B * c.T * 4, hipMemcpyDeviceToDevice, stream));
      }
      int rc = esm_gibbs_device(c.d_tok, c.B, c.T, c.d_idx, c.n_iters, c.P, &c.sp, c.lg, c.st);
      if (rc) return rc;
    }
    PG_HIP(hipStreamSynchronize(stream));
    chain_replay_ok = true;
    return fail(PG_ERR_HIP, "single-chain trunk: a device-wide barrier timed out; the device-pointer calls since the last "
                            "synchronisation were run again on the per-layer launches (their results are valid)");
  }
  return fail(PG_ERR_HIP, "single-chain trunk: a device-wide barrier timed out (is another persistent kernel sharing this GPU? "
                          "PGIBBS_CHAIN_TRUNK=0 selects the per-layer launches); the results of this call are invalid");
}

// strict precision mode GEMM: one bf16 MFMA GEMM over the K-concatenated split operands (engine.h) reproduces an
// fp32 x fp32 product to ~2^-17 relative (the dropped lo.lo term); fp32 accumulation, small terms first.
int Engine::dense3(const bf16_t* x3, const DenseW& W, float* out, int Mp, bool accumulate) {
  return timed(PC_GEMM, [&] { return launch_gemm_split3(stream, x3, W.w, W.b, out, Mp, W.N, W.K, W.N, accumulate ? EPI_F32_RESID : EPI_F32); });
}

// GEMM height for the B*P selected rows (pruned last layer, LM head) and for a forward of few token rows.  Up to 48 rows would
// take the weight-streaming kernel, whose k order differs from the tile kernels': fine for a few chains (its regime), but in a
// big batch the height is raised to one 64-row tile so that a shard and the whole batch see the same arithmetic.
int Engine::sel_gemm_rows(int64_t n_sel, int64_t Np) const {
  if (n_sel > 256) return (int)Np;
  const int r = round_up((int)n_sel, 16);
  return (batch_rows > 2048 && r < 64) ? 64 : r;
}

// The split operand rows are [lo | hi | hi] per 32 columns; the fused three-product kernel reads [lo | hi] only (gemm_w16.hip), the
// plain kernels over K' = 3K all three blocks.  A producer (LayerNorm, fc1's epilogue) skips the duplicate block -- a fifth of a
// LayerNorm's traffic, a third of fc1's stores -- exactly when its consumer is the fused kernel.
bool Engine::dense3_wants_dup(const DenseW& W, int Mp, bool gelu) const {
  static const int nodup = [] { const char* e = getenv("PGIBBS_SPLIT3_NODUP"); return e ? atoi(e) : 1; }();
  if (!nodup) return true;
  if (gelu && W.N % 256 == 0 && Mp % 256 == 0) return false;          // dense3_gelu's fused form: always the 16-wave kernel
  return !gemm_split3_fused(Mp, W.N, W.K, EPI_F32);
}

// strict fc1: ffn = split3(gelu(x3 . W^T + b)).  Fused into the GEMM's epilogue when the 16-wave kernel can take the shape
// (d_ffn a multiple of 256), else an fp32 GEMM followed by the GELU-and-split pass.  The fused epilogue evaluates the GELU
// with the degree-5 fit of log2 Phi(-|x|) that the bf16 mode uses (abs error 3.2e-6, below the 2^-17 relative error of the
// split products at |x| ~ 1; 9 instead of 22 issue slots per element in an epilogue that nothing overlaps), the pass with erff.
// Full-size strict logits against the oracle with this: ESM-1b 5.8e-4, MSA-1b 4.6e-4 (5.7e-4 / 4.0e-4 with erff).
int Engine::dense3_gelu(const bf16_t* x3, const DenseW& W, int Mp, const DenseW* next) {
  if (W.N % 256 == 0 && Mp % 256 == 0) {
    // the rows leave without their duplicate hi block when the projection that reads them (fc2) never looks at it
    const int epi = next && !dense3_wants_dup(*next, Mp) ? EPI_SPLIT2_GELU : EPI_SPLIT3_GELU;
    return timed(PC_GEMM, [&] { return launch_gemm_split3(stream, x3, W.w, W.b, ffn.as<bf16_t>(), Mp, W.N, W.K, 3 * W.N, epi); });
  }
  int rc = ffn_f32.ensure((size_t)Mp * W.N * 4, stream);       // fp32 intermediate of the unfused form only
  if (rc || (rc = dense3(x3, W, ffn_f32.as<float>(), Mp, false))) return rc;
  return timed(PC_LN, [&] { return launch_split3_bf16(stream, ffn_f32.as<float>(), ffn.as<bf16_t>(), Mp, W.N, 1.f, true, false); });
}

// A K-split sums in a different order than the one-pass kernels, so the decision must not depend on how a batch is sharded:
// it is taken on the forward's token rows M (`batch_rows`), also for the pruned last layer whose GEMMs see only the B*P selected
// rows -- a 32-chain shard of config 3 (8256 token rows, 800 selected) must give the same logits bit for bit as the whole
// 256-chain batch (6400 selected), and did not while this looked at the selected rows.
float* Engine::splitk_ws(int rows, int n, int64_t batch_rows) {
  if (rows > 2048 || batch_rows > 2048) return nullptr;
  const size_t need = (size_t)5 * round_up(rows, kRowPad) * n * 4;
  if (splitk.bytes < need && splitk.ensure(need, stream)) return nullptr;
  return splitk.as<float>();
}

int Engine::esm_trunk(const int32_t* d_tok, int B, int T, const int32_t* sel_idx, int P, int64_t n_sel, const int32_t* d_iter_) {
  const int d = cfg.d_model, f = cfg.d_ffn;
  const int64_t M = (int64_t)B * T;
  const int64_t Mp = round_up64(M, kRowPad);
  if (Mp > 0x7fffffff / 4) return fail(PG_ERR_INVALID, "too many tokens in one call");
  batch_rows = job_batch(B) * T;
  int rc;
  if (strict()) {
    if ((rc = x.ensure((size_t)Mp * d * 4, stream))) return rc;
    if ((rc = h.ensure((size_t)Mp * 3 * d * 2, stream))) return rc;        // [lo | hi | hi] rows
    if ((rc = qkv.ensure((size_t)Mp * 3 * d * 4, stream))) return rc;
    if ((rc = ctx.ensure((size_t)Mp * 3 * d * 2, stream))) return rc;
    if ((rc = ffn.ensure((size_t)Mp * 3 * f * 2, stream))) return rc;
    float* X = x.as<float>();
    float* QKVf = qkv.as<float>();
    const float eps = cfg.layer_norm_eps;
    const int Mi = (int)Mp;
    rc = timed(PC_EMBED, [&] {
      return OPS(launch_embed_ln, stream, d_tok, embed, pos, nullptr, ln_before.g, ln_before.b, X, M, T, d, cfg.pad_idx,
                             cfg.mask_idx, cfg.token_dropout, 0, eps, nullptr, nullptr, nullptr, embed_scale());
    });
    if (rc) return rc;
    const SeqLayout chain = {1, T, 0, 1};
    for (int l = 0; l < cfg.n_layers; ++l) {
      const EsmLayer& L = esm_layers[l];
      if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, X, L.ln1.g, L.ln1.b, h.as<bf16_t>(), M, d, eps, true, 0, 0, dense3_wants_dup(L.qkv, Mi)); }))) return rc;
      if ((rc = dense3(h.as<bf16_t>(), L.qkv, QKVf, Mi, false))) return rc;
      if ((rc = timed(PC_ATTN, [&] { return launch_attention_f32(stream, QKVf, ctx.as<bf16_t>(), dense3_wants_dup(L.out, Mi) ? d : -d, B, T, cfg.n_heads, 3 * d, 3 * d, d, 2 * d, chain, esm_pad_in_batch ? d_tok : nullptr, cfg.pad_idx, L.bias_kv32); }))) return rc;
      if ((rc = dense3(ctx.as<bf16_t>(), L.out, X, Mi, true))) return rc;
      if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, X, L.ln2.g, L.ln2.b, h.as<bf16_t>(), M, d, eps, true, 0, 0, dense3_wants_dup(L.fc1, Mi, true)); }))) return rc;
      if ((rc = dense3_gelu(h.as<bf16_t>(), L.fc1, Mi, &L.fc2))) return rc;
      if ((rc = dense3(ffn.as<bf16_t>(), L.fc2, X, Mi, true))) return rc;
    }
    return PG_OK;
  }
  if ((rc = x.ensure((size_t)Mp * d * 4, stream))) return rc;
  if ((rc = h.ensure((size_t)Mp * d * 2, stream))) return rc;
  if ((rc = qkv.ensure((size_t)Mp * 3 * d * 2, stream))) return rc;
  if ((rc = ctx.ensure((size_t)Mp * d * 2, stream))) return rc;
  if ((rc = ffn.ensure((size_t)Mp * f * 2, stream))) return rc;
  float* X = x.as<float>();
  bf16_t* Hh = h.as<bf16_t>();
  bf16_t* QKV = qkv.as<bf16_t>();
  bf16_t* CTX = ctx.as<bf16_t>();
  bf16_t* FFN = ffn.as<bf16_t>();
  const float eps = cfg.layer_norm_eps;
  // <= 256 rows: 64-row tiles, <= 48 rows the weight-streaming GEMM (buffers stay 256-padded) -- the latter not for a tiny shard
  // of a big job (sel_gemm_rows)
  const int Mi = sel_gemm_rows(M, Mp);
  // single chains (<= 32 token rows): LayerNorm is folded into the operand load of the weight-streaming QKV / fc1 GEMMs
  // (gemm_ln_skinny_kernel) -- two launches per layer less in the launch-bound regime
  const bool ln_in_gemm = gemm_ln_skinny_ok(Mi, 3 * d, d) && gemm_ln_skinny_ok(Mi, f, d);

  // embedding + emb_layer_norm_before, and in the same pass over the row the first layer's LayerNorm (its bf16 operand rows)
  rc = timed(PC_EMBED, [&] {
    return OPS(launch_embed_ln, stream, d_tok, embed, pos, nullptr, ln_before.g, ln_before.b, X, M, T, d, cfg.pad_idx,
                           cfg.mask_idx, cfg.token_dropout, 0, eps, ln_in_gemm ? nullptr : esm_layers[0].ln1.g,
                           ln_in_gemm ? nullptr : esm_layers[0].ln1.b, ln_in_gemm ? nullptr : Hh, embed_scale());
  });
  if (rc) return rc;
  // a single short chain (<= 32 token rows): every layer in ONE persistent launch (chain_trunk.hip; same bits as the per-layer
  // launches below).  With a selection it stops after the last layer's attention and the pruned tail below finishes on the
  // selected rows.
  int l_first = 0;
  bool attn_done = false;
  if (ln_in_gemm && chain_layers && !chain_disabled && !esm_pad_in_batch && T <= 32 && M <= Mi && OPS(chain_trunk_ok, Mi, d, f, cfg.n_heads)) {
    if ((rc = chain_sync.ensure(OPS(chain_trunk_sync_bytes), stream)) || (rc = chain_part.ensure(OPS(chain_trunk_part_bytes, Mi, d), stream))) return rc;
    PgChainTrunkArgs ca;
    ca.layers = chain_layers; ca.n_layers = cfg.n_layers; ca.partial_last = sel_idx ? 1 : 0; ca.B = B; ca.T = T;
    ca.x = X; ca.qkv = QKV; ca.ctx = CTX; ca.ffn = FFN; ca.part = chain_part.as<float>(); ca.sync = chain_sync.as<unsigned>();
    ca.err = chain_err; ca.eps = eps;
    rc = timed(PC_GEMM, [&] { return OPS(launch_chain_trunk, stream, ca, Mi, d); });
    if (rc == kChainTrunkUnfit) {              // the grid cannot be co-resident on this device: per-layer launches, for good
      chain_disabled = true;
      goto per_layer;
    }
    if (rc) return rc;
    {   // test hook: PGIBBS_CHAIN_TRUNK_FAULT=n makes the n-th persistent launch of the process report a barrier timeout
      static const int fault_at = [] { const char* e = getenv("PGIBBS_CHAIN_TRUNK_FAULT"); return e ? atoi(e) : 0; }();
      static int launches = 0;
      if (fault_at > 0 && ++launches == fault_at) *chain_err = 1;
    }
    if (!sel_idx) return PG_OK;
    l_first = cfg.n_layers - 1;
    attn_done = true;
  }
per_layer:
  for (int l = l_first; l < cfg.n_layers; ++l) {
    const EsmLayer& L = esm_layers[l];
    // Hh holds LN1(x): written by the previous layer's fc2 launch (or the LayerNorm kernel) -- see resid_gemm_ln
    if (attn_done) {
      // q | k | v and the attention of this (last) layer were part of the persistent launch
    } else if (ln_in_gemm) {
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_ln_skinny, stream, X, d, L.ln1.g, L.ln1.b, eps, L.qkv.w, L.qkv.b, QKV, Mi, 3 * d, d, d, 3 * d, EPI_BF16); }))) return rc;
    } else {
      if ((rc = timed(PC_GEMM_QKV, [&] { return OPS(launch_gemm_bf16, stream, Hh, L.qkv.w, L.qkv.b, QKV, Mi, 3 * d, d, d, d, 3 * d, EPI_BF16); }))) return rc;
    }
    if (!attn_done)
      if ((rc = timed(PC_ATTN, [&] { return OPS(launch_attention_bf16, stream, QKV, CTX, B, T, cfg.n_heads, 3 * d, d, d, 2 * d, esm_pad_in_batch ? d_tok : nullptr, cfg.pad_idx, L.bias_kv16); }))) return rc;
    if (sel_idx && l == cfg.n_layers - 1) {
      // last layer: only the selected rows are ever read again -> gather them and finish the layer on n_sel rows
      const int64_t Np = round_up64(n_sel, kRowPad);
      if ((rc = x_sel.ensure((size_t)Np * d * 4, stream)) || (rc = ctx_sel.ensure((size_t)Np * d * 2, stream)) ||
          (rc = h_sel.ensure((size_t)Np * d * 2, stream)) || (rc = ffn_sel.ensure((size_t)Np * f * 2, stream))) return rc;
      float* XS = x_sel.as<float>();
      const int Ni = sel_gemm_rows(n_sel, Np);
      rc = timed(PC_HEAD, [&] {
        int r2 = launch_gather_rows(stream, X, XS, sel_idx, nullptr, P, T, n_sel, d * 4, d_iter_);
        if (r2) return r2;
        return launch_gather_rows(stream, CTX, ctx_sel.as<bf16_t>(), sel_idx, nullptr, P, T, n_sel, d * 2, d_iter_);
      });
      if (rc) return rc;
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, ctx_sel.as<bf16_t>(), L.out.w, L.out.b, XS, Ni, d, d, d, d, d, EPI_F32_RESID); }))) return rc;
      if (ln_in_gemm && gemm_ln_skinny_ok(Ni, f, d)) {
        if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_ln_skinny, stream, XS, d, L.ln2.g, L.ln2.b, eps, L.fc1.w, L.fc1.b, ffn_sel.as<bf16_t>(), Ni, f, d, d, f, EPI_BF16_GELU); }))) return rc;
      } else {
        if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, XS, L.ln2.g, L.ln2.b, h_sel.as<bf16_t>(), n_sel, d, eps); }))) return rc;
        if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, h_sel.as<bf16_t>(), L.fc1.w, L.fc1.b, ffn_sel.as<bf16_t>(), Ni, f, d, d, d, f, EPI_BF16_GELU); }))) return rc;
      }
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, ffn_sel.as<bf16_t>(), L.fc2.w, L.fc2.b, XS, Ni, d, f, f, f, d, EPI_F32_RESID, splitk_ws(Ni, d, batch_rows), splitk.bytes); }))) return rc;
      break;
    }
    if (ln_in_gemm) {
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, CTX, L.out.w, L.out.b, X, Mi, d, d, d, d, d, EPI_F32_RESID); }))) return rc;
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_ln_skinny, stream, X, d, L.ln2.g, L.ln2.b, eps, L.fc1.w, L.fc1.b, FFN, Mi, f, d, d, f, EPI_BF16_GELU); }))) return rc;
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, FFN, L.fc2.w, L.fc2.b, X, Mi, d, f, f, f, d, EPI_F32_RESID, splitk_ws(Mi, d, batch_rows), splitk.bytes, (int)M); }))) return rc;
      continue;
    }
    if ((rc = resid_gemm_ln(CTX, L.out, X, Mi, M, d, L.ln2, Hh, nullptr, 0, PC_GEMM_OUT))) return rc;   // x += out_proj(ctx); h = LN2(x)
    if ((rc = timed(PC_GEMM_FC1, [&] { return OPS(launch_gemm_bf16, stream, Hh, L.fc1.w, L.fc1.b, FFN, Mi, f, d, d, d, f, EPI_BF16_GELU, nullptr, 0, (int)M); }))) return rc;
    if (l + 1 < cfg.n_layers) {                                                                    // x += fc2(ffn); h = LN1 of the next layer
      if ((rc = resid_gemm_ln(FFN, L.fc2, X, Mi, M, f, esm_layers[l + 1].ln1, Hh, splitk_ws(Mi, d, batch_rows), splitk.bytes, PC_GEMM_FC2))) return rc;
    } else {
      if ((rc = timed(PC_GEMM_FC2, [&] { return OPS(launch_gemm_bf16, stream, FFN, L.fc2.w, L.fc2.b, X, Mi, d, f, f, f, d, EPI_F32_RESID, splitk_ws(Mi, d, batch_rows), splitk.bytes, (int)M); }))) return rc;
    }
  }
  return PG_OK;
}

// x[M_rows][d] += a[M_rows][K] W^T + b, then h = LayerNorm(x; ln) as the next GEMM's bf16 operand: the residual GEMM the dispatch
// picks, followed by the LayerNorm kernel at its HBM roofline.  (Normalising inside the GEMM was built in round 3, bit-identical
// and slower; it left the library in round 4: tools/probes/gemm_ln_fused.hip.)
int Engine::resid_gemm_ln(const bf16_t* a, const DenseW& W, float* x, int M_rows, int64_t M_real, int lda, const LnW& ln, bf16_t* h,
                          float* ws, size_t ws_bytes, int prof_class, int colmajor_R, int colmajor_C) {
  const int d = W.N, K = W.K;
  // d_model = 768, K <= 1024 (ESM-MSA-1b's attention out-projections) on big batches: one launch whose tiles span whole rows and
  // normalise them in the epilogue (gemm_rowln.hip; same bits as the two launches below, so the choice is free)
  static const int64_t rowln_min = [] { const char* e = getenv("PGIBBS_ROWLN_MIN_ROWS"); return e ? atoll(e) : 16384LL; }();
  if (!strict() && M_real >= rowln_min && M_real <= M_rows && OPS(gemm_rowln_ok, M_rows, d, K))
    return timed(prof_class, [&] { return OPS(launch_gemm_rowln, stream, a, W.w, W.b, x, ln.g, ln.b, h, (int)M_real, M_rows, K, lda, K, cfg.layer_norm_eps, colmajor_R, colmajor_C); });
  int rc = timed(prof_class, [&] { return OPS(launch_gemm_bf16, stream, a, W.w, W.b, x, M_rows, d, K, lda, K, d, EPI_F32_RESID, ws, ws_bytes, (int)M_real); });
  if (rc) return rc;
  return timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, x, ln.g, ln.b, h, M_real, d, cfg.layer_norm_eps, false, colmajor_R, colmajor_C); });
}

// LM head evaluated ONLY at the selected rows: emb_layer_norm_after -> dense -> GELU ->
// LayerNorm -> tied decoder + bias.  d_idx == nullptr: every one of the n_sel rows of x in order.
int Engine::head(const int32_t* d_idx_, const int32_t* d_row_map, int P, int width, int64_t n_sel, float* d_logits,
                 const float* x_src) {
  if (!x_src) x_src = x.as<float>();
  const int d = cfg.d_model, V = cfg.vocab;
  const int64_t Np = round_up64(n_sel, kRowPad);
  int rc;
  if (esm1()) {
    // ESM-1: logits = x embed_out^T + embed_out_bias on the selected rows of the residual stream (no final LayerNorm, no dense);
    // the decoder kernel with its LayerNorm switched off (gamma == nullptr), fp32 throughout in every precision mode
    const float* rows = x_src;
    if (d_idx_) {
      if ((rc = sel_g.ensure((size_t)Np * d * 4, stream))) return rc;
      if ((rc = launch_gather_rows(stream, x_src, sel_g.as<float>(), d_idx_, d_row_map, P, width, n_sel, d * 4))) return rc;
      rows = sel_g.as<float>();
    }
    return timed(PC_HEAD, [&] { return launch_lm_tail(stream, rows, nullptr, nullptr, embed_out, head_bias, d_logits, n_sel, d, V, cfg.layer_norm_eps); });
  }
  if ((rc = sel_h.ensure((size_t)Np * d * 2, stream))) return rc;
  if ((rc = sel_g.ensure((size_t)Np * d * 4, stream))) return rc;
  const float eps = cfg.layer_norm_eps;
  if (strict()) {
    if ((rc = sel_h.ensure((size_t)Np * 3 * d * 2, stream))) return rc;
    if ((rc = OPS(launch_gather_ln_bf16, stream, x_src, d_idx_, d_row_map, P, width, ln_after.g, ln_after.b,
                                    sel_h.as<bf16_t>(), n_sel, d, eps, true))) return rc;
    if ((rc = dense3(sel_h.as<bf16_t>(), head_dense, sel_g.as<float>(), (int)Np, false))) return rc;
    if ((rc = launch_gelu_f32(stream, sel_g.as<float>(), Np * d))) return rc;
    return launch_lm_tail(stream, sel_g.as<float>(), head_ln.g, head_ln.b, embed, head_bias, d_logits, n_sel, d, V, eps);
  }
  return timed(PC_HEAD, [&] {
    int r = OPS(launch_gather_ln_bf16, stream, x_src, d_idx_, d_row_map, P, width, ln_after.g, ln_after.b,
                                  sel_h.as<bf16_t>(), n_sel, d, eps);
    if (r) return r;
    r = OPS(launch_gemm_bf16, stream, sel_h.as<bf16_t>(), head_dense.w, head_dense.b, sel_g.as<float>(),
                         sel_gemm_rows(n_sel, Np), d, d, d, d, d, EPI_F32_GELU);
    if (r) return r;
    return launch_lm_tail(stream, sel_g.as<float>(), head_ln.g, head_ln.b, embed, head_bias, d_logits, n_sel, d, V, eps);
  });
}

int Engine::esm_gibbs_device(int32_t* d_tok, int B, int T, const int32_t* d_idx_, int n_iters, int P,
                             const pg_sample_params* sp, float* d_samp_logits_, int32_t* d_samp_tok_) {
  if (cfg.arch != PG_ARCH_ESM1B && cfg.arch != PG_ARCH_ESM1) return fail(PG_ERR_INVALID, "engine was not built for an ESM-1b / ESM-1 architecture");
  if (B < 0 || T < 1 || P < 0 || n_iters < 0) return fail(PG_ERR_INVALID, "gibbs: negative size");
  if (T > cfg.max_positions) return fail(PG_ERR_INVALID, "sequence longer than the learned position table");
  if (B == 0 || n_iters == 0) return PG_OK;
  const int V = cfg.vocab;
  const int64_t n_sel_rows = B;              // one selected token row per chain
  const int64_t n_draws = (int64_t)B * P;
  int rc;
  if (!d_samp_logits_ && (rc = logits.ensure((size_t)(n_draws > 0 ? n_draws : 1) * V * 4, stream))) return rc;
  static const int prune = [] { const char* e = getenv("PGIBBS_PRUNE_LAST"); return e ? atoi(e) : 1; }();
  static const int use_graph = [] { const char* e = getenv("PGIBBS_GRAPH"); return e ? atoi(e) : 1; }();
  const bool pruned = prune && !strict() && P > 0 && n_draws * 2 < (int64_t)B * T;

  // one Gibbs iteration; with dit != nullptr every iteration-dependent quantity is derived on the device from *dit
  auto iteration = [&](int it, const int32_t* dit) -> int {
    const int32_t* idx_it = dit ? d_idx_ : d_idx_ + (size_t)it * n_draws;
    int r;
    if (sp->mask && P > 0)
      if ((r = timed(PC_SAMPLE, [&] { return launch_mask_scatter(stream, d_tok, T, idx_it, nullptr, n_sel_rows, P, sp->mask_idx, dit); }))) return r;
    if ((r = pruned ? esm_trunk(d_tok, B, T, idx_it, P, n_draws, dit) : esm_trunk(d_tok, B, T))) return r;
    if (P == 0) return PG_OK;
    float* lg = d_samp_logits_ ? d_samp_logits_ + (size_t)it * n_draws * V : logits.as<float>();
    if ((r = pruned ? head(nullptr, nullptr, 1, 1, n_draws, lg, x_sel.as<float>()) : head(idx_it, nullptr, P, T, n_draws, lg))) return r;
    int32_t* st = d_samp_tok_ ? d_samp_tok_ + (size_t)it * n_draws : nullptr;
    return timed(PC_SAMPLE, [&] { return launch_sample_writeback(stream, d_tok, T, lg, V, 1, idx_it, nullptr, n_sel_rows, P, sp, dit ? 0 : it, st, dit, range_err); });
  };

  // Launch-bound regime (few tokens: ~270 launches of a few microseconds each): capture ONE iteration as a hipGraph and
  // replay it.  Needs the pruned path (no per-iteration pointers besides the idx table), no per-iteration outputs, no
  // event profiling.  The iteration number and the sampling parameters (seed, burn-in, top-k ...) live in a device-side state
  // block, so the SAME graph is replayed by every later call of the same shape on the same buffers -- a generate() per
  // sequence (BASELINE config 1) captures once, not once per call.  Before a capture iteration 0 runs eagerly (it sizes every
  // workspace buffer: no allocation inside the capture).
  const bool graphable = use_graph && pruned && !prof.on && !d_samp_logits_ && !d_samp_tok_ && (int64_t)B * T <= 4096;
  std::vector<uint8_t> key(2 * sizeof(int32_t) + 8 * sizeof(int64_t));
  if (graphable) {
    const int32_t flags[2] = {sp->mask, sp->mask_idx};      // baked into the captured launches (the scatter and its argument)
    memcpy(key.data(), flags, sizeof(flags));
    // job_items picks kernels (sel_gemm_rows, splitk_ws): a graph captured under another job size must not be replayed
    const int64_t dims[8] = {(int64_t)(uintptr_t)d_tok, (int64_t)(uintptr_t)d_idx_, B, T, P, (int64_t)(uintptr_t)stream,
                             (int64_t)(g_alloc_epoch * 2 + (esm_pad_in_batch ? 1 : 0)), job_items};
    memcpy(key.data() + sizeof(flags), dims, sizeof(dims));
  }
  const bool reuse = graphable && graph_exec && key == graph_key;
  if (!graphable || (!reuse && n_iters < 3)) {
    for (int it = 0; it < n_iters; ++it)
      if ((rc = iteration(it, nullptr))) return rc;
    return PG_OK;
  }
  int first = 0;
  if (!reuse) {
    if ((rc = d_iter.ensure(graph_state_bytes(), stream))) return rc;
    if ((rc = iteration(0, nullptr))) return rc;
    first = 1;
    {   // a buffer grown by iteration 0 moved the epoch: the key must describe the state the graph is captured in
      const int64_t ep = (int64_t)(g_alloc_epoch * 2 + (esm_pad_in_batch ? 1 : 0));
      memcpy(key.data() + 2 * sizeof(int32_t) + 6 * sizeof(int64_t), &ep, sizeof(ep));
    }
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    hipGraph_t graph = nullptr;
    PG_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    rc = iteration(0, d_iter.as<int32_t>());
    if (!rc) rc = launch_iter_counter(stream, d_iter.as<int32_t>(), false, 0);
    hipError_t ce = hipStreamEndCapture(stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) return fail(PG_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    ce = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ce != hipSuccess) { graph_exec = nullptr; return fail(PG_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ce)); }
    graph_key = key;
    ++stat_graph_captures;
  }
  // replay: the counter selects the idx slice, the Philox iteration word and the burn-in flag; the parameters are this call's
  if ((rc = launch_graph_state(stream, d_iter.as<int32_t>(), first, sp))) return rc;
  for (int it = first; it < n_iters; ++it) PG_HIP(hipGraphLaunch(graph_exec, stream));
  stat_graph_replays += n_iters - first;
  return PG_OK;
}

// ------------------------------------------------------------------------------------------------
// ESM-MSA-1b forward: tokens[B][R][C] -> x[B*R*C][d]
// ------------------------------------------------------------------------------------------------
int Engine::msa_trunk(const int32_t* d_tok, int B, int R, int C, const int32_t* sel_idx, const int32_t* sel_row_map, int P,
                      int64_t n_sel) {
  const int d = cfg.d_model, f = cfg.d_ffn, H = cfg.n_heads;
  const int64_t M = (int64_t)B * R * C;
  const int64_t Mp = round_up64(M, kRowPad);
  if (Mp > 0x7fffffff / 4) return fail(PG_ERR_INVALID, "too many tokens in one call");
  batch_rows = job_batch(B) * R * C;
  int rc;
  if ((rc = x.ensure((size_t)Mp * d * 4, stream))) return rc;
  if ((rc = h.ensure((size_t)Mp * d * 2, stream))) return rc;
  if ((rc = qkv.ensure((size_t)Mp * 3 * d * 2, stream))) return rc;
  if ((rc = ctx.ensure((size_t)Mp * d * 2, stream))) return rc;
  if ((rc = ffn.ensure((size_t)Mp * f * 2, stream))) return rc;
  float* X = x.as<float>();
  bf16_t* Hh = h.as<bf16_t>();
  bf16_t* QKV = qkv.as<bf16_t>();
  bf16_t* CTX = ctx.as<bf16_t>();
  bf16_t* FFN = ffn.as<bf16_t>();
  const float eps = cfg.layer_norm_eps;
  const int Mi = (int)Mp;
  const float row_scale = 0.125f / sqrtf((float)R);      // dh^-0.5 / sqrt(R): depends on R, applied to the scores
  // A batch that holds <pad> (a ragged list of MSAs padded to one tensor by the unmasked log_likelihood_batch, esm_msa_sampler.py:341,
  // 416-431; set by the host-token entry points): fair-esm's padding semantics -- tied row attention with q zeroed at <pad>
  // positions and row 0's <pad> columns filled with -10000 (the scores-through-scratch kernels), column attention with <pad> key
  // rows filled with -10000.  R, and with it the 1/sqrt(R) of the row attention, is the padded row count, as in the reference.
  const int32_t* pad_tok = esm_pad_in_batch ? d_tok : nullptr;
  if (pad_tok && precision == PG_PREC_F16)
    return fail(PG_ERR_UNSUPPORTED, "fp16 precision mode: batches with <pad> (ragged MSA lists) take the scores-through-scratch row "
                                    "attention, which exists for bf16 operands only -- use precision bf16 or fp32");
  if (strict()) {
    if ((rc = h.ensure((size_t)Mp * 3 * d * 2, stream)) || (rc = ctx.ensure((size_t)Mp * 3 * d * 2, stream))) return rc;   // [lo | hi | hi] rows
    if ((rc = qkv.ensure((size_t)Mp * 3 * d * 4, stream))) return rc;
    if ((rc = ffn.ensure((size_t)Mp * 3 * f * 2, stream))) return rc;
    if ((rc = scores.ensure((size_t)B * H * C * msa_row_scores_ld(C) * 4, stream))) return rc;
    float* Xs = x.as<float>();
    float* QKVf = qkv.as<float>();
    bf16_t *H3 = h.as<bf16_t>(), *C3 = ctx.as<bf16_t>();
    const float eps2 = cfg.layer_norm_eps;
    const int Mi2 = (int)Mp;
    rc = timed(PC_EMBED, [&] {
      return OPS(launch_embed_ln, stream, d_tok, embed, pos, msa_pos, ln_before.g, ln_before.b, Xs, M, C, d, cfg.pad_idx,
                             cfg.mask_idx, 0, R, eps2);
    });
    if (rc) return rc;
    const SeqLayout colS = {C, R * C, 1, C};
    for (int l = 0; l < cfg.n_layers; ++l) {
      const MsaLayer& L = msa_layers[l];
      if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, Xs, L.ln_row.g, L.ln_row.b, H3, M, d, eps2, true, 0, 0, dense3_wants_dup(L.row_qkv, Mi2)); }))) return rc;
      if ((rc = dense3(H3, L.row_qkv, QKVf, Mi2, false))) return rc;
      if ((rc = timed(PC_ATTN, [&] { return launch_msa_row_attention_f32(stream, QKVf, scores.as<float>(), C3, dense3_wants_dup(L.row_out, Mi2) ? d : -d, B, R, C, H, 3 * d, 3 * d, d, 2 * d, row_scale, pad_tok, cfg.pad_idx); }))) return rc;
      if ((rc = dense3(C3, L.row_out, Xs, Mi2, true))) return rc;
      if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, Xs, L.ln_col.g, L.ln_col.b, H3, M, d, eps2, true, 0, 0, dense3_wants_dup(L.col_qkv, Mi2)); }))) return rc;
      if ((rc = dense3(H3, L.col_qkv, QKVf, Mi2, false))) return rc;
      if ((rc = timed(PC_ATTN, [&] { return launch_attention_f32(stream, QKVf, C3, dense3_wants_dup(L.col_out, Mi2) ? d : -d, (int64_t)B * C, R, H, 3 * d, 3 * d, d, 2 * d, colS, pad_tok, cfg.pad_idx); }))) return rc;
      if ((rc = dense3(C3, L.col_out, Xs, Mi2, true))) return rc;
      if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, Xs, L.ln_ffn.g, L.ln_ffn.b, H3, M, d, eps2, true, 0, 0, dense3_wants_dup(L.fc1, Mi2, true)); }))) return rc;
      if ((rc = dense3_gelu(H3, L.fc1, Mi2, &L.fc2))) return rc;
      if ((rc = dense3(ffn.as<bf16_t>(), L.fc2, Xs, Mi2, true))) return rc;
    }
    return PG_OK;
  }

  // embedding + emb_layer_norm_before, and in the same pass over the row the first layer's row-attention LayerNorm
  rc = timed(PC_EMBED, [&] {
    return OPS(launch_embed_ln, stream, d_tok, embed, pos, msa_pos, ln_before.g, ln_before.b, X, M, C, d, cfg.pad_idx,
                           cfg.mask_idx, 0, R, eps, msa_layers[0].ln_row.g, msa_layers[0].ln_row.b, Hh);
  });
  if (rc) return rc;
  const SeqLayout col = {C, R * C, 1, C};                   // column c of msa b: rows (b*R + r)*C + c
  // Hh always holds the LayerNorm the next projection reads: written by the embedding pass / the LayerNorm after a residual GEMM
  for (int l = 0; l < cfg.n_layers; ++l) {
    const MsaLayer& L = msa_layers[l];
    // tied row attention
    if ((rc = timed(PC_GEMM_QKV, [&] { return OPS(launch_gemm_bf16, stream, Hh, L.row_qkv.w, L.row_qkv.b, QKV, Mi, 3 * d, d, d, d, 3 * d, EPI_BF16); }))) return rc;
    if (C <= 576 && !pad_tok) {
      float* part = nullptr;
      // few workgroups: give the kernel scratch for its split-R mode (decided on the job's batch: the split changes the
      // order of the sum over alignment rows, and a shard must compute what the whole batch would)
      const size_t need = msa_row_split_scratch_bytes(B, R, C, H, (int)(job_batch(B) * H));
      if (need) {
        // the host entry points chunk a batch of templates so that this fits (api.hip); a device-pointer caller gets the reason
        if (need > ((size_t)1 << 31)) return fail(PG_ERR_INVALID, "too many MSAs in one call for the row-split attention scratch: use smaller batches");
        if ((rc = scores.ensure(need, stream))) return rc;           // the allocator's own error (out of memory)
        part = scores.as<float>();
      }
      if ((rc = timed(PC_ATTN, [&] { return OPS(launch_msa_row_attention_bf16, stream, QKV, CTX, B, R, C, H, 3 * d, d, d, 2 * d, row_scale, part, part ? scores.bytes : 0, (int)(job_batch(B) * H)); }))) return rc;
    } else {
      // alignments wider than the MFMA row-attention kernel's register budget: fp32 scores through a scratch buffer
      if (precision == PG_PREC_F16)
        return fail(PG_ERR_UNSUPPORTED, "fp16 precision mode: alignments wider than 576 columns take the split-bf16 row attention, "
                                        "which exists for bf16 operands only -- use precision bf16 or fp32");
      if ((rc = scratch.ensure((size_t)Mp * 3 * d * 4, stream)) || (rc = scores.ensure((size_t)B * H * C * msa_row_scores_ld(C) * 4, stream))) return rc;
      rc = timed(PC_ATTN, [&] {
        int r2 = launch_bf16_to_f32(stream, QKV, scratch.as<float>(), (int64_t)M * 3 * d);
        if (r2) return r2;
        return launch_msa_row_attention_f32(stream, scratch.as<float>(), scores.as<float>(), CTX, 0, B, R, C, H,
                                            3 * d, d, d, 2 * d, row_scale, pad_tok, cfg.pad_idx);
      });
      if (rc) return rc;
    }
    // column block: fused QKV projection + attention when the depth allows (gemm_colattn.hip: the LayerNorm writes its rows in
    // column-major token order, q / k / v never leave the CU; bit-identical context), else projection + attention kernel
    const bool col_fused = !pad_tok && colattn_ok(R, d, H) && L.col_qkv_hm.w;
    if ((rc = resid_gemm_ln(CTX, L.row_out, X, Mi, M, d, L.ln_col, Hh, nullptr, 0, PC_GEMM_OUT, col_fused ? R : 0, col_fused ? C : 0))) return rc;   // x += row_out(ctx); h = LN_col(x)
    if (col_fused) {
      if ((rc = timed(PC_GEMM_QKV, [&] { return OPS(launch_gemm_colattn, stream, Hh, L.col_qkv_hm.w, L.col_qkv_hm.b, CTX, B, R, C, H, d, d); }))) return rc;
    } else {
      // column attention (q pre-scaled by dh^-0.5 in the weights)
      if ((rc = timed(PC_GEMM_QKV, [&] { return OPS(launch_gemm_bf16, stream, Hh, L.col_qkv.w, L.col_qkv.b, QKV, Mi, 3 * d, d, d, d, 3 * d, EPI_BF16); }))) return rc;
      if ((rc = timed(PC_ATTN, [&] { return OPS(launch_attention_seq_bf16, stream, QKV, CTX, (int64_t)B * C, R, H, 3 * d, d, d, 2 * d, col, pad_tok, cfg.pad_idx, nullptr); }))) return rc;
    }
    if (sel_idx && l == cfg.n_layers - 1) {
      // last layer: nothing but the selected rows is read again -> finish column out-projection and FFN on n_sel rows
      const int64_t Np = round_up64(n_sel, kRowPad);
      if ((rc = x_sel.ensure((size_t)Np * d * 4, stream)) || (rc = ctx_sel.ensure((size_t)Np * d * 2, stream)) ||
          (rc = h_sel.ensure((size_t)Np * d * 2, stream)) || (rc = ffn_sel.ensure((size_t)Np * f * 2, stream))) return rc;
      float* XS = x_sel.as<float>();
      const int Ni = (int)Np;
      rc = timed(PC_HEAD, [&] {
        int r2 = launch_gather_rows(stream, X, XS, sel_idx, sel_row_map, P, C, n_sel, d * 4);
        if (r2) return r2;
        return launch_gather_rows(stream, CTX, ctx_sel.as<bf16_t>(), sel_idx, sel_row_map, P, C, n_sel, d * 2);
      });
      if (rc) return rc;
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, ctx_sel.as<bf16_t>(), L.col_out.w, L.col_out.b, XS, Ni, d, d, d, d, d, EPI_F32_RESID); }))) return rc;
      if ((rc = timed(PC_LN, [&] { return OPS(launch_layernorm_bf16, stream, XS, L.ln_ffn.g, L.ln_ffn.b, h_sel.as<bf16_t>(), n_sel, d, eps); }))) return rc;
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, h_sel.as<bf16_t>(), L.fc1.w, L.fc1.b, ffn_sel.as<bf16_t>(), Ni, f, d, d, d, f, EPI_BF16_GELU); }))) return rc;
      if ((rc = timed(PC_GEMM, [&] { return OPS(launch_gemm_bf16, stream, ffn_sel.as<bf16_t>(), L.fc2.w, L.fc2.b, XS, Ni, d, f, f, f, d, EPI_F32_RESID, splitk_ws(Ni, d, batch_rows), splitk.bytes); }))) return rc;
      break;
    }
    if ((rc = resid_gemm_ln(CTX, L.col_out, X, Mi, M, d, L.ln_ffn, Hh, nullptr, 0, PC_GEMM_OUT))) return rc;                 // x += col_out(ctx); h = LN_ffn(x)
    // feed forward
    if ((rc = timed(PC_GEMM_FC1, [&] { return OPS(launch_gemm_bf16, stream, Hh, L.fc1.w, L.fc1.b, FFN, Mi, f, d, d, d, f, EPI_BF16_GELU, nullptr, 0, (int)M); }))) return rc;
    if (l + 1 < cfg.n_layers) {                                                                      // x += fc2(ffn); h = LN_row of the next layer
      if ((rc = resid_gemm_ln(FFN, L.fc2, X, Mi, M, f, msa_layers[l + 1].ln_row, Hh, splitk_ws(Mi, d, batch_rows), splitk.bytes, PC_GEMM_FC2))) return rc;
    } else {
      if ((rc = timed(PC_GEMM_FC2, [&] { return OPS(launch_gemm_bf16, stream, FFN, L.fc2.w, L.fc2.b, X, Mi, d, f, f, f, d, EPI_F32_RESID, splitk_ws(Mi, d, batch_rows), splitk.bytes, (int)M); }))) return rc;
    }
  }
  return PG_OK;
}

static int check_msa_shape(const pg_model_config& cfg, int B, int R, int C) {
  if (cfg.arch != PG_ARCH_MSA1B) return fail(PG_ERR_INVALID, "engine was not built for the MSA-1b architecture");
  if (B < 0 || R < 1 || C < 1) return fail(PG_ERR_INVALID, "bad MSA shape");
  if (R > cfg.max_msa_rows) return fail(PG_ERR_INVALID, "MSA has more rows than msa_position_embedding");
  if (C > cfg.max_positions) return fail(PG_ERR_INVALID, "alignment longer than the learned position table");
  return PG_OK;
}

int Engine::msa_gibbs_device(int32_t* d_tok, int B, int R, int C, const int32_t* d_idx_, int n_iters, int P,
                             const pg_sample_params* sp, float* d_samp_logits_, int32_t* d_samp_tok_) {
  int rc = check_msa_shape(cfg, B, R, C);
  if (rc) return rc;
  if (P < 0 || n_iters < 0) return fail(PG_ERR_INVALID, "gibbs: negative size");
  if (B == 0 || n_iters == 0) return PG_OK;
  const int V = cfg.vocab;
  const int64_t n_sel_rows = (int64_t)B * R;         // every row of every MSA draws its own P positions
  const int64_t n_draws = n_sel_rows * P;
  if (!d_samp_logits_ && (rc = logits.ensure((size_t)(n_draws > 0 ? n_draws : 1) * V * 4, stream))) return rc;
  for (int it = 0; it < n_iters; ++it) {
    const int32_t* idx_it = d_idx_ + (size_t)it * n_draws;
    if (sp->mask && P > 0)
      if ((rc = timed(PC_SAMPLE, [&] { return launch_mask_scatter(stream, d_tok, C, idx_it, nullptr, n_sel_rows, P, sp->mask_idx); }))) return rc;
    static const int prune = [] { const char* e = getenv("PGIBBS_PRUNE_LAST"); return e ? atoi(e) : 1; }();
    const bool pruned = prune && !strict() && P > 0 && n_draws * 2 < n_sel_rows * C;
    if ((rc = pruned ? msa_trunk(d_tok, B, R, C, idx_it, nullptr, P, n_draws) : msa_trunk(d_tok, B, R, C))) return rc;
    if (P == 0) continue;
    float* lg = d_samp_logits_ ? d_samp_logits_ + (size_t)it * n_draws * V : logits.as<float>();
    if ((rc = pruned ? head(nullptr, nullptr, 1, 1, n_draws, lg, x_sel.as<float>()) : head(idx_it, nullptr, P, C, n_draws, lg))) return rc;
    int32_t* st = d_samp_tok_ ? d_samp_tok_ + (size_t)it * n_draws : nullptr;
    if ((rc = timed(PC_SAMPLE, [&] { return launch_sample_writeback(stream, d_tok, C, lg, V, 1, idx_it, nullptr, n_sel_rows, P, sp, it, st, nullptr, range_err); }))) return rc;
  }
  return PG_OK;
}

// generate_single for B templates of equal shape at once ("batch=32 templates").  Template b is token block
// b; step s masks row mask_row of every template and samples row target_row of template b at d_step_idx[s][b][0..P_max) with
// template b's own sampling parameters sp[b] (its own Philox key: the reference draws a fresh torch seed per call).  Every
// order-changing kernel choice is taken as if a template were alone (order_items = 1), so template b's logits and tokens are
// bit-identical with a B = 1 call on it -- and therefore independent of how templates are batched or sharded over GPUs.
int Engine::msa_single_device(int32_t* d_tok, int B, int R, int C, int mask_row, int target_row, const int32_t* d_step_idx,
                              const int32_t* step_sample_flag_host, int n_steps, int P_max, const pg_sample_params* sp,
                              float* d_samp_logits_, int32_t* d_samp_tok_) {
  int rc = check_msa_shape(cfg, B, R, C);
  if (rc) return rc;
  if (mask_row < 0 || mask_row >= R || target_row < 0 || target_row >= R) return fail(PG_ERR_INVALID, "row index out of range");
  if (P_max < 0 || n_steps < 0) return fail(PG_ERR_INVALID, "negative size");
  if (n_steps == 0 || B == 0) return PG_OK;
  if (B > 1 && (int64_t)R * C <= 2048) {
    // few-token regime: K-splits and the weight-streaming GEMM are picked by the local shape there, so a
    // batch would not reproduce the single calls -- run the templates one after the other (launch-bound either way)
    for (int b = 0; b < B; ++b) {
      // template b's steps, contiguous [n_steps][P_max]
      if ((rc = tmp_idx.ensure((size_t)(n_steps * (size_t)P_max + 1) * 4, stream))) return rc;
      for (int s2 = 0; s2 < n_steps && P_max > 0; ++s2)
        PG_HIP(hipMemcpyAsync(tmp_idx.as<int32_t>() + (size_t)s2 * P_max, d_step_idx + ((size_t)s2 * B + b) * P_max, (size_t)P_max * 4,
                              hipMemcpyDeviceToDevice, stream));
      // per-template outputs land in scratch-free strided form: sample into temporaries, then scatter
      float* lg_b = nullptr;
      int32_t* st_b = nullptr;
      if (d_samp_logits_ || d_samp_tok_) {
        if ((rc = tmp_out.ensure((size_t)n_steps * (P_max > 0 ? P_max : 1) * cfg.vocab * 4 + (size_t)n_steps * (P_max > 0 ? P_max : 1) * 4, stream))) return rc;
        lg_b = d_samp_logits_ ? tmp_out.as<float>() : nullptr;
        st_b = d_samp_tok_ ? (int32_t*)(tmp_out.as<float>() + (size_t)n_steps * (P_max > 0 ? P_max : 1) * cfg.vocab) : nullptr;
      }
      if ((rc = msa_single_device(d_tok + (size_t)b * R * C, 1, R, C, mask_row, target_row, tmp_idx.as<int32_t>(), step_sample_flag_host,
                                  n_steps, P_max, sp + b, lg_b, st_b))) return rc;
      for (int s2 = 0; s2 < n_steps && P_max > 0; ++s2) {
        if (lg_b) PG_HIP(hipMemcpyAsync(d_samp_logits_ + ((size_t)s2 * B + b) * P_max * cfg.vocab, lg_b + (size_t)s2 * P_max * cfg.vocab,
                                        (size_t)P_max * cfg.vocab * 4, hipMemcpyDeviceToDevice, stream));
        if (st_b) PG_HIP(hipMemcpyAsync(d_samp_tok_ + ((size_t)s2 * B + b) * P_max, st_b + (size_t)s2 * P_max, (size_t)P_max * 4,
                                        hipMemcpyDeviceToDevice, stream));
      }
    }
    return PG_OK;
  }
  const int V = cfg.vocab;
  if ((rc = d_rowmap.ensure((size_t)2 * B * 4, stream))) return rc;
  std::vector<int32_t> maps((size_t)2 * B);
  for (int b = 0; b < B; ++b) {
    maps[b] = b * R + mask_row;
    maps[B + b] = b * R + target_row;
  }
  PG_HIP(hipMemcpyAsync(d_rowmap.p, maps.data(), maps.size() * 4, hipMemcpyHostToDevice, stream));
  PG_HIP(hipStreamSynchronize(stream));     // `maps` is a local buffer
  const int32_t* d_mask_map = d_rowmap.as<int32_t>();
  const int32_t* d_tgt_map = d_rowmap.as<int32_t>() + B;
  const int64_t n_draws = (int64_t)B * P_max;
  if (!d_samp_logits_ && (rc = logits.ensure((size_t)(n_draws > 0 ? n_draws : 1) * V * 4, stream))) return rc;
  struct OrderGuard { Engine& e; int64_t prev; ~OrderGuard() { e.order_items = prev; } } guard{*this, order_items};
  order_items = 1;
  for (int s = 0; s < n_steps; ++s) {
    const int32_t* idx_s = d_step_idx + (size_t)s * n_draws;
    if (P_max > 0)   // generate_single always masks (esm_msa_sampler.py:133), row -1 regardless of the target row
      if ((rc = timed(PC_SAMPLE, [&] { return launch_mask_scatter(stream, d_tok, C, idx_s, d_mask_map, B, P_max, sp->mask_idx); }))) return rc;
    static const int prune = [] { const char* e = getenv("PGIBBS_PRUNE_LAST"); return e ? atoi(e) : 1; }();
    const bool pruned = prune && !strict() && P_max > 0 && (int64_t)P_max * 2 < (int64_t)R * C;
    if ((rc = pruned ? msa_trunk(d_tok, B, R, C, idx_s, d_tgt_map, P_max, n_draws) : msa_trunk(d_tok, B, R, C))) return rc;
    if (P_max == 0) continue;
    float* lg = d_samp_logits_ ? d_samp_logits_ + (size_t)s * n_draws * V : logits.as<float>();
    if ((rc = pruned ? head(nullptr, nullptr, 1, 1, n_draws, lg, x_sel.as<float>()) : head(idx_s, d_tgt_map, P_max, C, n_draws, lg))) return rc;
    int32_t* st = d_samp_tok_ ? d_samp_tok_ + (size_t)s * n_draws : nullptr;
    for (int b = 0; b < B; ++b) {           // one draw launch per template: its own Philox key, Philox row id = row_id_base
      pg_sample_params p = sp[b];
      p.burnin = step_sample_flag_host[s] ? 0x7fffffff : 0;   // sample=(pass_num < burn_in), esm_msa_sampler.py:143
      if ((rc = timed(PC_SAMPLE, [&] { return launch_sample_writeback(stream, d_tok, C, lg + (size_t)b * P_max * V, V, 1, idx_s + (size_t)b * P_max,
                                                                      d_tgt_map + b, 1, P_max, &p, s, st ? st + (size_t)b * P_max : nullptr, nullptr, range_err); }))) return rc;
    }
  }
  return PG_OK;
}

}  // namespace pg
