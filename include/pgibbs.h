/* pgibbs.h -- C ABI of the MI355X-native Gibbs-sampling engine for masked protein LMs.
 *
 * Plain pointers and sizes only (no torch / C++ types), so any FFI (ctypes, cffi, cgo, JNI ...)
 * can bind it.  Every entry point names the reference interface it replaces; paths are relative
 * to the reference tree (seanrjohnson/protein_gibbs_sampler, `pgen` 0.2.3).
 *
 * Conventions
 *   - every function returning int returns PG_OK (0) or a PG_ERR_* code; pg_last_error() gives the
 *     message of the last failure on the calling thread;
 *   - the caller owns every host buffer; the engine owns its device memory; one engine per device;
 *     calls on one engine are serialized by the caller (the reference is single-threaded too);
 *   - no callbacks; safe to call with the Python GIL released (ctypes default);
 *   - token buffers are int32 row-major; logits are fp32 row-major;
 *   - "*_device" entry points take pointers that are already resident in HBM on the engine's
 *     device and run on the engine's stream (pg_engine_set_stream).
 */
#ifndef PGIBBS_H
#define PGIBBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_ERR_INVALID 1   /* bad argument / shape */
#define PG_ERR_HIP 2       /* HIP runtime failure (message has hipGetErrorString) */
#define PG_ERR_NO_DEVICE 3 /* no MI355X visible -- the product never falls back to a CPU path */
#define PG_ERR_WEIGHTS 4   /* missing / mis-shaped tensor in the state dict */
#define PG_ERR_UNSUPPORTED 5
#define PG_ERR_RANGE 6     /* non-finite logits reached the draw / log-probability / output stage: a 16-bit tensor of the forward left
                              the fp16 range in PG_PREC_F16 (or the weights hold NaN / inf).  Host-buffer entry points report it
                              BEFORE they overwrite the caller's buffers, so the call can be repeated on a PG_PREC_BF16 engine;
                              device-pointer entry points report it at pg_engine_synchronize */

const char* pg_version(void);
const char* pg_last_error(void);
/* number of HIP devices visible (0 when there is none or the runtime cannot initialise) */
int pg_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Host: CPython-exact Mersenne Twister.
 * Replaces the reference's use of the *global* Python RNG on the hot path:
 *   random.sample(indexes, num_positions)   src/pgen/esm_sampler.py:245, esm_msa_sampler.py:277
 *   random.choices(seed_seq, k=batch_size)  src/pgen/esm_sampler.py:112
 *   random.shuffle(positions)               src/pgen/esm_msa_sampler.py:129
 * The Python layer moves random.getstate() in, generates the whole position table natively, and
 * moves the advanced state back with random.setstate(), so position selection is bit-exact and
 * the interpreter's RNG is left exactly where the reference would have left it.
 * ------------------------------------------------------------------------------------------ */
typedef struct pg_pyrandom pg_pyrandom;
pg_pyrandom* pg_pyrandom_create(void);
void pg_pyrandom_destroy(pg_pyrandom*);
/* random.seed(int): key = little-endian 32-bit words of abs(n) (at least one word) */
int pg_pyrandom_seed(pg_pyrandom*, const uint32_t* key_words, int n_words);
int pg_pyrandom_setstate(pg_pyrandom*, const uint32_t* mt624, int index);
int pg_pyrandom_getstate(const pg_pyrandom*, uint32_t* mt624, int* index);
uint32_t pg_pyrandom_getrandbits32(pg_pyrandom*, int k /* 1..32 */);
double pg_pyrandom_random(pg_pyrandom*);
/* random.sample(population, k): out[k] = chosen population values */
int pg_pyrandom_sample(pg_pyrandom*, const int32_t* population, int n, int k, int32_t* out);
/* n_rows successive random.sample calls (one per chain / MSA row), out[n_rows][k] */
int pg_pyrandom_sample_table(pg_pyrandom*, const int32_t* population, int n, int k, int64_t n_rows, int32_t* out);
int pg_pyrandom_shuffle(pg_pyrandom*, int32_t* x, int n);
/* random.choices(range(n), k=k) (no weights): out[k] = chosen indices */
int pg_pyrandom_choices(pg_pyrandom*, int n, int k, int32_t* out);

/* ------------------------------------------------------------------------------------------
 * Engine: holds the weights of one masked LM on one MI355X and runs the Gibbs hot path.
 * Replaces the `model.model` attribute of the reference's plug-in object
 * (src/pgen/models.py:59-88; call sites esm_sampler.py:223, esm_msa_sampler.py:136,236) plus the
 * per-position Python loop around it (esm_sampler.py:209-234, esm_msa_sampler.py:128-145,221-248).
 * ------------------------------------------------------------------------------------------ */
typedef struct pg_engine pg_engine;

#define PG_ARCH_ESM1B 1 /* fair-esm ProteinBertModel, "ESM-1b" architecture (ESM-1b, ESM-1v) */
#define PG_ARCH_MSA1B 2 /* fair-esm MSATransformer (esm_msa1b_t12_100M_UR50S) */
#define PG_ARCH_ESM1 3  /* fair-esm ProteinBertModel, "ESM-1" architecture (esm1_t6_43M / t12_85M / t34_670M_UR50S: the models behind
                           pgen.models.ESM6 / ESM12 / ESM34, /root/reference/src/pgen/models.py:69-82): embed_scale sqrt(d), sinusoidal
                           positions (supplied as the `embed_positions.weight` table), no emb_layer_norm_before / after, one extra
                           attention key / value per layer (`layers.i.self_attn.bias_k` / `bias_v`, d_model values each), untied output
                           projection `embed_out.weight` [V][d] + `embed_out.bias` [V], no LM-head dense / LayerNorm, no token
                           dropout; vocabulary of 35 (<cls> = 32, <mask> = 33).  Runs through the ESM-1b entry points. */

#define PG_PREC_BF16 0 /* bf16 MFMA operands, fp32 accumulate, fp32 residual stream (throughput mode) */
#define PG_PREC_F16 2  /* the throughput mode with IEEE fp16 operands instead of bf16 (same kernels, same MFMA rate; 3 more mantissa
                          bits: logit error 8x smaller, profiles/r04_rounding_ablation.txt).  No saturation: a 16-bit tensor of the
                          forward beyond +-65504 becomes inf, the next LayerNorm / softmax turns it into NaN logits, and the call
                          returns PG_ERR_RANGE instead of drawing from them -- use PG_PREC_BF16 for such checkpoints (the Python
                          layer's precision="auto" does that by itself: weights.py / engine.py). */
#define PG_PREC_FP32 1 /* strict parity mode: every matrix product (projections, q.k^T, P.v) as three bf16 MFMA products on
                          (hi, lo) splits of both operands (lo.hi + hi.lo + hi.hi, fp32 accumulate); fp32 softmax, LayerNorm
                          and residual stream; ~3x slower, logits within 1e-3 of the fp32 oracle.  The FFN's GELU is
                          evaluated as relu(x) - |x| 2^p(|x|) with a degree-5 fit p of log2 Phi(-t) (max abs error 3.2e-6,
                          below the split products' own ~2^-17 relative error) when d_ffn is a multiple of 256 -- both
                          models -- and with erff otherwise */

typedef struct {
  int32_t arch;
  int32_t vocab;         /* 33 */
  int32_t d_model;       /* 1280 (ESM-1b) / 768 (MSA-1b); multiple of 128 */
  int32_t n_layers;      /* 33 / 12 */
  int32_t n_heads;       /* d_model / 64: head dim is 64 in both models */
  int32_t d_ffn;         /* 5120 / 3072; multiple of 128 */
  int32_t max_positions; /* learned position table has max_positions + pad_idx + 1 rows */
  int32_t pad_idx, mask_idx, cls_idx, eos_idx;
  int32_t token_dropout; /* 1 for ESM-1b, 0 for MSA-1b */
  int32_t max_msa_rows;  /* rows of msa_position_embedding (1024), 0 for ESM-1b */
  float layer_norm_eps;  /* 1e-5 */
} pg_model_config;

/* one fp32 tensor of a fair-esm state dict, by its fair-esm key.
 * The decoder is tied: logits = LN(gelu(dense(x))) . embed_tokens^T + lm_head.bias.  fair-esm zeroes the <mask> row of
 * embed_tokens when it loads an ESM-1b checkpoint (token dropout), and the Python loader (weights.load_fair_esm_checkpoint) does
 * the same -- so after a real ESM-1b checkpoint load logit[<mask>] = lm_head.bias[<mask>] only.  The samplers never draw <mask>
 * (not in valid_idx); log-likelihoods normalise over all V logits including that bias-only term, as fair-esm does.  Tensors
 * handed to pg_engine_create directly are used as given. */
typedef struct {
  const char* name;
  const float* data;
  int64_t numel;
} pg_tensor;

int pg_engine_create(const pg_model_config* cfg, const pg_tensor* tensors, int n_tensors, int device_ordinal,
                     int precision, pg_engine** out);
void pg_engine_destroy(pg_engine*);
/* run on a caller-provided hipStream_t (NULL = the engine's own stream) */
int pg_engine_set_stream(pg_engine*, void* hip_stream);
int pg_engine_synchronize(pg_engine*);
int pg_engine_device(const pg_engine*);
/* Multi-GPU jobs (contiguous blocks of chains / MSAs per GPU): the number of batch items (chains, MSAs) of the
 * WHOLE job this engine's calls are a shard of; 0 (default) = every call is a whole job.  Kernel choices that change the
 * order of a floating-point sum (K-split GEMMs and the weight-streaming GEMM of the few-chain regime, the row-split form of
 * the tied row attention) are then taken on the job's size instead of the shard's, so that every shard computes bit for bit
 * what a single engine computes on the whole batch (jobs with more than 2048 token rows; smaller jobs run in the few-chain
 * regime whose kernels are picked by the local shape).  No effect on results of a call that is a whole job. */
int pg_engine_set_job_items(pg_engine*, int64_t job_items);
/* engine counters: "graph_captures" / "graph_replays" = hipGraph captures and replayed iterations of the launch-bound
 * (few-token) Gibbs loop, which replays ONE captured iteration for every call of the same shape (tests assert the path taken) */
int pg_engine_get_stat(pg_engine*, const char* name, int64_t* value);

/* Sampling parameters == the kwargs of generate() that reach generate_step
 * (src/pgen/esm_sampler.py:8-45, 227-232). */
typedef struct {
  int32_t mask;        /* scatter <mask> at the targets before the forward (esm_sampler.py:220-221) */
  int32_t mask_idx;    /* alphabet.mask_idx */
  int32_t top_k;       /* as passed by the caller; the sample/top_k rule of :32-38 is applied inside */
  int32_t burnin;      /* iterations ii < burnin draw from the full distribution; INT32_MAX = inf */
  float temperature;   /* NaN == None (no division) */
  int32_t n_valid;     /* <= 32 */
  int32_t valid_idx[32]; /* sampler.valid_aa_idx (esm_sampler.py:82, esm_msa_sampler.py:65) */
  uint64_t rng_seed;   /* pg_draw v1 Philox key */
  uint32_t rng_stream; /* Philox counter word 3 */
  uint32_t row_id_base;/* global id of row 0 (Philox counter word 0 = row_id_base + row): results are
                          identical for any sharding of the chains over GPUs */
  int32_t iter_base;   /* Philox counter word 1 = iter_base + iteration */
} pg_sample_params;

/* ---- ESM-1b -------------------------------------------------------------------------------
 * pg_esm_forward_logits: model.model(tokens)["logits"] (esm_sampler.py:223): tokens[B,T] -> logits[B,T,V].
 * pg_esm_gibbs_run: the whole `for ii in range(num_iters)` loop of ESM_sampler.generate
 *   (esm_sampler.py:209-234) for one batch: per iteration mask scatter -> forward -> restrict /
 *   temperature / top-k / categorical -> write-back, all on the device with no host round trip.
 *   target_idx[n_iters][B][P] are the (1-based token) positions chosen on the host; entries < 0 are
 *   skipped (ragged lists); positions >= T are PG_ERR_INVALID (the *_device variants skip them); bit 30 set = position already written by a later duplicate in the same
 *   row (sampled but not written back: keeps the reference's sequential last-write-wins result).
 *   Optional outputs (NULL to skip): sampled_logits[n_iters][B][P][V], sampled_tokens[n_iters][B][P].
 */
int pg_esm_forward_logits(pg_engine*, const int32_t* tokens, int B, int T, float* logits_out);
int pg_esm_gibbs_run(pg_engine*, int32_t* tokens_inout, int B, int T, const int32_t* target_idx, int n_iters, int P,
                     const pg_sample_params* params, float* sampled_logits, int32_t* sampled_tokens);
/* same, every pointer device-resident (tokens stay in HBM; used by bench.py and multi-GPU drivers).  Asynchronous: returns after
 * enqueueing on the engine's stream; every buffer (incl. d_target_idx) must stay valid until pg_engine_synchronize.  Single short
 * chains (<= 32 token rows) may run on a persistent launch whose device-wide barriers can time out when the GPU is shared; such
 * calls are logged with a snapshot of their token rows, and the next pg_esm_gibbs_run_device / pg_engine_synchronize that sees
 * the timeout restores the rows and runs the logged calls again on the per-layer launches (same results, one warning).
 * Consequence for the caller: between two synchronisations nothing but pg_*_device calls of this engine may write a token buffer
 * that was handed to such a call -- the replay restores the snapshot taken at the first logged call and re-runs only those. */
int pg_esm_gibbs_run_device(pg_engine*, int32_t* d_tokens_inout, int B, int T, const int32_t* d_target_idx,
                            int n_iters, int P, const pg_sample_params* params, float* d_sampled_logits,
                            int32_t* d_sampled_tokens);

/* ---- ESM-MSA-1b ---------------------------------------------------------------------------
 * tokens[B][R][C] (C = L + 1: <cls> then L columns, no <eos>).
 * pg_msa_gibbs_run: loop of ESM_MSA_sampler.generate (esm_msa_sampler.py:221-248):
 *   target_idx[n_iters][B][R][P].
 * pg_msa_gibbs_single_run: loop of ESM_MSA_sampler.generate_single (esm_msa_sampler.py:128-145), B = 1:
 *   n_steps forwards; step s masks row `mask_row` and samples row `target_row` at
 *   step_idx[s][P_max] (entries < 0 = padding of the shorter partitions); sample flag per step.
 */
int pg_msa_forward_logits(pg_engine*, const int32_t* tokens, int B, int R, int C, float* logits_out);
int pg_msa_gibbs_run(pg_engine*, int32_t* tokens_inout, int B, int R, int C, const int32_t* target_idx, int n_iters,
                     int P, const pg_sample_params* params, float* sampled_logits, int32_t* sampled_tokens);
int pg_msa_gibbs_run_device(pg_engine*, int32_t* d_tokens_inout, int B, int R, int C, const int32_t* d_target_idx,
                            int n_iters, int P, const pg_sample_params* params, float* d_sampled_logits,
                            int32_t* d_sampled_tokens);
int pg_msa_gibbs_single_run(pg_engine*, int32_t* tokens_inout, int R, int C, int mask_row, int target_row,
                            const int32_t* step_idx, const int32_t* step_sample_flag, int n_steps, int P_max,
                            const pg_sample_params* params, float* sampled_logits, int32_t* sampled_tokens);
/* pg_msa_gibbs_single_batch_run: B calls of generate_single on B MSAs ("templates") of equal shape R x C in ONE pass -- the inner
 *   loop of pgen_msa_revised (src/pgen/pgen_msa_revised.py:107-115: one generate_single per template and per requested sequence;
 *   BASELINE config 5 "batch=32 templates").  tokens[B][R][C]; step_idx[n_steps][B][P_max] (template b's partition of ITS OWN
 *   shuffled positions, < 0 = padding); step_sample_flag[n_steps] (shared: same passes / burn_in); params[B] -- template b draws
 *   with params[b] (the reference draws one torch seed per call), Philox row id = params[b].row_id_base.
 *   sampled_logits[n_steps][B][P_max][V], sampled_tokens[n_steps][B][P_max] optional.
 *   Template b's result is bit-identical with pg_msa_gibbs_single_run on it alone (kernel choices that change a summation order
 *   are taken per template), hence independent of how templates are batched or sharded over GPUs. */
int pg_msa_gibbs_single_batch_run(pg_engine*, int32_t* tokens_inout, int B, int R, int C, int mask_row, int target_row,
                                  const int32_t* step_idx, const int32_t* step_sample_flag, int n_steps, int P_max,
                                  const pg_sample_params* params, float* sampled_logits, int32_t* sampled_tokens);

/* ---- masked log-likelihood scoring ----------------------------------------------------------------
 * The forward + log_softmax + gather of log_likelihood_batch (src/pgen/esm_sampler.py:336-348,355-362;
 * src/pgen/esm_msa_sampler.py:398-410,421-431).  Sample s scores token row row_of[s] (ESM: the chain index;
 * MSA: b*R + target_index) at positions idx[s][P] (entries < 0 are skipped and yield 0) against targets[s][P];
 * out[s][P] = log_softmax(logits[row, pos, :])[target].  The LM head runs only at the scored rows.
 * pg_logprob_gather_device: the same gather on a caller's device-resident full logits[n_rows][width][V]. */
int pg_esm_forward_logprobs(pg_engine*, const int32_t* tokens, int B, int T, const int32_t* row_of, const int32_t* idx,
                            const int32_t* targets, int n_sel, int P, float* out);
int pg_msa_forward_logprobs(pg_engine*, const int32_t* tokens, int B, int R, int C, const int32_t* row_of,
                            const int32_t* idx, const int32_t* targets, int n_sel, int P, float* out);
int pg_logprob_gather_device(void* stream, const float* d_logits, int64_t n_rows, int width, int V, const int32_t* d_idx,
                             const int32_t* d_row_map, const int32_t* d_targets, int64_t n_sel, int P, float* d_out);

/* ---- stand-alone data-parallel ends of the iteration --------------------------------------
 * For plug-in models whose forward is not this engine (the reference accepts any object with
 * .model/.alphabet/.batch_converter, esm_sampler.py:54-58): logits come from the caller's model,
 * mask scatter and the draw run here.  All pointers device-resident; `stream` is a hipStream_t.
 *   tokens[n_rows][width] int32; idx[n_sel][P]; row_map[n_sel] (NULL: row r = r) maps a selected row to
 *   its token row; logits[n_rows][width][V] fp32 (full model output).
 * pg_mask_scatter_device  == mask_target_indexes      (esm_sampler.py:259-262, esm_msa_sampler.py:255-264)
 * pg_sample_writeback_device == generate_step + write (esm_sampler.py:225-234)
 */
int pg_mask_scatter_device(void* stream, int32_t* d_tokens, int64_t n_rows, int width, const int32_t* d_idx,
                           const int32_t* d_row_map, int64_t n_sel, int P, int mask_idx);
int pg_sample_writeback_device(void* stream, int32_t* d_tokens, int64_t n_rows, int width, const float* d_logits,
                               int V, const int32_t* d_idx, const int32_t* d_row_map, int64_t n_sel, int P,
                               const pg_sample_params* params, int iteration, int32_t* d_sampled_tokens);

/* ---- multi-GPU tail: the one collective of a sharded job -------------------------------------------
 * Chains / MSAs / templates are independent, so a job shards as contiguous blocks over the GPUs of a node with NO data-path
 * collective; what the reference untokenises at the end of a batch -- the token buffer of ALL chains,
 * src/pgen/esm_sampler.py:236-239, esm_msa_sampler.py:250-253 -- is rebuilt from the shards by one RCCL all-gather over xGMI.
 * One process per GPU, one communicator per process.  librccl.so is opened at run time (PG_ERR_UNSUPPORTED when absent).
 *   pg_comm_unique_id   rank 0 fills id_out[PG_COMM_ID_BYTES] (ncclGetUniqueId) and hands it to the other ranks out of band
 *                       (environment, file, MPI, a torch.distributed store ...)
 *   pg_comm_create      every rank, same id: joins the communicator on `device_ordinal` (ncclCommInitRank; collective)
 *   pg_gather_tokens    d_local[rows][width] int32 of every rank -> d_out[sum(counts)][width] on every rank, rank-major (= chain
 *                       order of contiguous shards).  counts[world] = rows of each rank (NULL: every rank has `rows`); equal
 *                       counts gather straight into d_out, ragged ones through padded blocks.  Runs on `hip_stream` (a
 *                       hipStream_t, NULL = the null stream), asynchronously: synchronise the stream before reading d_out on
 *                       the host.  Order it after the engine's work (pg_engine_synchronize, or pass the engine's stream). */
#define PG_COMM_ID_BYTES 128
typedef struct pg_comm pg_comm;
int pg_comm_unique_id(void* id_out);
int pg_comm_create(int rank, int world, const void* unique_id, int device_ordinal, pg_comm** out);
void pg_comm_destroy(pg_comm*);
int pg_comm_rank(const pg_comm*);
int pg_comm_world(const pg_comm*);
int pg_gather_tokens(pg_comm*, void* hip_stream, const int32_t* d_local, int64_t rows, int width, const int64_t* counts,
                     int32_t* d_out);
/* Rehearsal of pg_gather_tokens without GPUs (tests only): the SAME bookkeeping -- which form, block size, scratch layout, per-rank
 * pack offsets -- on HOST buffers, with the all-gather injected by the caller: allgather(ctx, send, recv, n) must fill recv with
 * the `world` ranks' n int32 values in rank order and return 0.  One call per simulated rank (e.g. one thread each); world sizes
 * 2 ... 8 with ragged and empty shards run in the CPU suite this way (tests/test_gather_rehearsal.py).  pg_dbg_gather_plan reports
 * the plan itself: out7 = {equal form?, nothing to move?, rows of a padded block, bytes per row, bytes per block, scratch bytes,
 * bytes of the gathered output}, out_off_bytes[world] = byte offset of every rank's rows in the output (may be NULL). */
typedef int (*pg_allgather_fn)(void* ctx, const void* send, void* recv, size_t n_int32);
int pg_dbg_gather_tokens_host(int rank, int world, const int32_t* local, int64_t rows, int width, const int64_t* counts,
                              int force_padded, pg_allgather_fn allgather, void* ctx, int32_t* out);
int pg_dbg_gather_plan(int rank, int world, int64_t rows, int width, const int64_t* counts, int force_padded, int64_t* out7,
                       int64_t* out_off_bytes);

/* ---- measurement ---------------------------------------------------------------------------
 * HIP-event timing of kernel classes on the engine's stream (bench.py's roofline figure).
 * class names: "gemm", "attention", "layernorm", "embed", "head", "sample".  */
int pg_prof_enable(pg_engine*, int on);
int pg_prof_reset(pg_engine*);
int pg_prof_get(pg_engine*, const char* kernel_class, double* total_ms, int64_t* launches);
/* the kernels the GEMM dispatch picked for the profiled launches of a class ("gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2",
 * "gemm_other", "head"): distinct labels such as "pp192 220t" (220 tiles of 192 x 256) or "tile64 400t x4k" (64 x 64 tiles,
 * 4 K-splits), separated by " | ", into buf. */
int pg_prof_get_kernels(pg_engine*, const char* kernel_class, char* buf, int buf_bytes);

/* ---- kernel-level debug entry points (parity tests call individual kernels through these) --- */
/* out[M][N] = x[M][K] @ w[N][K]^T + bias (fp32 host buffers; computed in `precision`); epi: 0 none, 1 gelu,
 * 2 residual (out is read too: out += x w^T + bias), 3 bf16 output, 4 bf16 output + gelu.  PG_PREC_FP32 (the engine's
 * strict-mode projection: one GEMM over K-concatenated split-bf16 operands) takes epi 0, 2 and 5 = fc1's fused epilogue
 * (GELU, then the split operand rows fc2 reads; N a multiple of 256; out = hi + lo of those rows); PG_PREC_F16 = the bf16
 * path's kernels with fp16 operands (epi 0-4) */
int pg_dbg_gemm(int device, int precision, const float* x, const float* w, const float* bias, float* out, int M, int N,
                int K, int epi);
/* times `iters` back-to-back launches of the GEMM on device-resident random bf16 operands (HIP events; M a multiple of 16,
 * of 64 above 256); variant 1 =
 * lockstep kernel, 2 = ping-pong kernel; epi: 0 bf16 out, 1 bf16+gelu, 2 fp32 residual, 3 fp32, 4 fp32+gelu */
int pg_dbg_gemm_bench(int device, int M, int N, int K, int epi, int variant, int iters, double* avg_ms);
/* round 6: gemm_rowln.hip (out-projection of d_model = 768 with the next LayerNorm in its epilogue) against the two launches it
 * replaces: ms[0] fused, ms[1] its main loop alone, ms[2] four half-steps + epilogue, ms[3] residual GEMM (256-column tiles),
 * ms[4] LayerNorm kernel; max_diff 0 = x and h bit-identical between the two paths. */
int pg_dbg_rowln_bench(int device, int M, int K, int iters, double* ms, double* max_diff);
/* round-5 ablation: QKV projection + attention of B sequences of T in {32, 64, 128, 256} tokens as ONE launch (the fused
 * projection-attention kernel of the MSA column block, one head per 256 x 192 tile) against the two launches of the ESM-1b path;
 * ms[0] fused, ms[1] projection, ms[2] attention (HIP events, `iters` launches each); max_diff = max |fused - unfused| context */
int pg_dbg_qkv_attention_bench(int device, int B, int T, int H, int iters, double* ms, double* max_diff);
/* y = LayerNorm(x[M][d]) * gamma + beta */
int pg_dbg_layernorm(int device, const float* x, const float* gamma, const float* beta, float* y, int M, int d,
                     float eps);
/* softmax(q k^T) v per (b, h); q already scaled; qkv[B][T][3*H*64] fp32 -> ctx[B][T][H*64]; PG_PREC_BF16 or PG_PREC_FP32
 * (split-bf16 MFMA kernel, output = hi + lo of the operand rows it writes) or PG_PREC_F16 (the bf16 kernel with fp16 operands) */
int pg_dbg_attention(int device, int precision, const float* qkv, float* ctx, int B, int T, int H);

/* MSA attention blocks: qkv[B][R][C][3*H*64] fp32 -> ctx[B][R][C][H*64]; which = 0 tied row attention (scores * scale),
 * 1 column attention (q pre-scaled); 2 / 3 = the same two with the strict precision mode's kernels; 4 / 5 = 0 / 1 with fp16
 * operands */
int pg_dbg_msa_attention(int device, int which, const float* qkv, float* ctx, int B, int R, int C, int H, float scale);

#ifdef __cplusplus
}
#endif
#endif /* PGIBBS_H */
