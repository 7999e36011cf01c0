// The two integer/index ends of a Gibbs iteration, as data-parallel kernels:
//
//   mask_scatter_kernel      == mask_target_indexes           /root/reference/src/pgen/esm_sampler.py:259-262,
//                               (+ _single)                    esm_msa_sampler.py:255-264
//   sample_writeback_kernel  == generate_step + write-back    esm_sampler.py:8-45, 225-234;
//                                                              esm_msa_sampler.py:138-145, 238-248
//
// In the reference both are Python loops issuing B*P scalar tensor writes (6400 per iteration at
// config 2).  All P draws of an iteration read the same logits (no re-forward between positions), so
// they are independent: one thread per (row, slot).  A row of 33 logits is 132 B; the whole job is a
// few hundred KB -> latency-bound, one launch each.
//
// The draw is "pg_draw v1" (oracle/draw.py): every float op below is a separately rounded IEEE
// binary32 op -- this file is compiled with -ffp-contract=off -- so the kernel and the CPU oracle
// agree bit-for-bit given the same logits.
#include "../../include/pgibbs.h"
#include "kernels.h"

namespace pg {

#pragma clang fp contract(off)

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o0 = c0;
}

// exp(x), x <= 0, reproducible (oracle/draw.py::pg_exp)
__device__ __forceinline__ float pg_exp(float x) {
  const float z = x * 1.44269502162933349609375f;  // float32(log2 e)
  if (z < -126.0f) return 0.0f;
  const float n = floorf(z);
  const float f = z - n;
  float p = 0x1.b4d1dep-13f;
  p = p * f; p = p + 0x1.4ca4cep-10f;
  p = p * f; p = p + 0x1.3c4a8ap-7f;
  p = p * f; p = p + 0x1.c69f6ap-5f;
  p = p * f; p = p + 0x1.ebfc40p-3f;
  p = p * f; p = p + 0x1.62e430p-1f;
  p = p * f; p = p + 1.0f;
  const int bits = __float_as_int(p) + ((int)n << 23);
  return __int_as_float(bits);
}

constexpr int kGraphArgsOffset = 4;      // int32 words: d_iter[0] = iteration, SampleArgs from byte 16 on

struct SampleArgs {
  int32_t top_k;
  int32_t sample;       // 1: draw from all valid tokens regardless of top_k
  float temperature;
  int32_t use_temp;
  int32_t n_valid;
  int32_t valid_idx[32];
  uint32_t seed_lo, seed_hi, stream, row_id_base, iter;
  int32_t burnin;             // used with a device-side iteration counter: sample = (it < burnin)
};

// logits addressing: compact [n_sel*P][V] (engine path: LM head evaluated only at the sampled rows),
// or full [n_rows][width][V] (plug-in models).
// One WAVE per draw (round 4; it was one thread per draw -- a 32 x 32 rank loop and 32 exponentials in a single lane, 42 us
// however few draws there were, 3 % of a config-1 iteration): lane j holds valid entry j.  Every floating-point operation and its
// order is the one-thread form's (the cumulative sums are accumulated in rank order, one after the other, in a value all lanes
// hold), so the draws are the same bit for bit.
__global__ __launch_bounds__(256) void sample_writeback_kernel(int32_t* __restrict__ tokens, int width,
                                                              const float* __restrict__ logits, int V, int compact,
                                                              const int32_t* __restrict__ idx,
                                                              const int32_t* __restrict__ row_map, int64_t n_sel, int P,
                                                              SampleArgs a, int32_t* __restrict__ sampled_tokens,
                                                              const int32_t* __restrict__ d_iter, unsigned* nonfinite) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n_sel * P) return;   // wave-uniform
  if (d_iter) {               // graph replay: the iteration number AND the sampling parameters live on the device, so one
    const int it = *d_iter;   // captured graph serves every iteration of every call with this shape (launch_graph_state)
    a = *(const SampleArgs*)(d_iter + kGraphArgsOffset);
    idx += (size_t)it * n_sel * P;
    a.iter += (uint32_t)it;
    a.sample = it < a.burnin ? 1 : 0;
  }
  const int64_t s = i / P;
  const int slot = (int)(i - s * P);
  const int raw = idx[i];
  if (raw < 0) {
    if (sampled_tokens && lane == 0) sampled_tokens[i] = -1;
    return;
  }
  const int pos = raw & 0x3fffffff;
  if (pos >= width) {          // out-of-range position (the host entry points reject these; *_device callers are guarded here)
    if (sampled_tokens && lane == 0) sampled_tokens[i] = -1;
    return;
  }
  const int64_t trow = row_map ? (int64_t)row_map[s] : s;
  const float* row = compact ? logits + (size_t)i * V : logits + ((size_t)trow * width + pos) * V;

  const int nv = a.n_valid;
  int k = a.top_k;
  if (a.sample || k <= 0 || k > nv) k = nv;
  // this lane's entry of the valid list (static indices into the by-value struct: no private-memory copy)
  int my_valid = 0;
#pragma unroll
  for (int m = 0; m < 32; ++m)
    if (m == lane) my_valid = a.valid_idx[m];
  const bool mine = lane < nv;
  float v = 0.f;
  if (mine) {
    v = row[my_valid];
    // a non-finite logit (an overflowed fp16 operand upstream: inf -> NaN through the next LayerNorm) is reported, not sampled
    // from silently: the engine's entry points return PG_ERR_RANGE (Engine::range_check)
    if (nonfinite && !(fabsf(v) <= 3.0e38f)) *nonfinite = 1u;
    if (a.use_temp) v = v / a.temperature;
  }
  // stable descending rank of every valid entry
  int rank = 0;
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    if (m < nv) {                                      // wave-uniform
      const float vm = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), m));
      rank += (vm > v) || (vm == v && m < lane);
    }
  }
  // lane r receives the value and the valid-list position of the entry with rank r (rank is a permutation of 0..nv-1)
  const int dst = (mine ? rank : lane) * 4;            // idle lanes send to themselves
  const float ev = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, v)));
  const int ei = __builtin_amdgcn_ds_permute(dst, lane);
  const float top = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), 0));
  const float e = lane < k ? pg_exp(ev - top) : 0.f;
  float acc = 0.f, cum = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    if (r < k) {                                       // wave-uniform
      const float er = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e), r));
      acc = (r == 0) ? er : acc + er;
      if (lane == r) cum = acc;
    }
  }
  uint32_t w0;
  philox4x32_10(a.row_id_base + (uint32_t)s, a.iter, (uint32_t)slot, a.stream, a.seed_lo, a.seed_hi, w0);
  const float u = (float)(w0 >> 8) * 0x1.0p-24f;
  const float t = u * acc;
  const unsigned long long hit = __ballot(lane < k && cum > t);
  const int jsel = hit ? (int)__builtin_ctzll(hit) : k - 1;
  const int pick = __builtin_amdgcn_readlane(ei, jsel);
  const int tok = __builtin_amdgcn_readlane(my_valid, pick);
  if (lane == 0) {
    if (sampled_tokens) sampled_tokens[i] = tok;
    if (!(raw & 0x40000000)) tokens[trow * width + pos] = tok;
  }
}

__global__ __launch_bounds__(256) void mask_scatter_kernel(int32_t* __restrict__ tokens, int width,
                                                          const int32_t* __restrict__ idx,
                                                          const int32_t* __restrict__ row_map, int64_t n_sel, int P,
                                                          int mask_idx, const int32_t* __restrict__ d_iter) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_sel * P) return;
  if (d_iter) idx += (size_t)(*d_iter) * n_sel * P;
  const int raw = idx[i];
  if (raw < 0 || (raw & 0x3fffffff) >= width) return;
  const int64_t s = i / P;
  const int64_t trow = row_map ? (int64_t)row_map[s] : s;
  tokens[trow * width + (raw & 0x3fffffff)] = mask_idx;
}

// log p(target | context) at the selected rows: log_softmax over the FULL vocabulary then gather
// (== torch.log_softmax(logits, -1)[..., target], /root/reference/src/pgen/esm_sampler.py:340-345,
//  esm_msa_sampler.py:403-410).  One thread per selected entry; V <= 64 floats per row.
__global__ __launch_bounds__(256) void logprob_gather_kernel(const float* __restrict__ logits, int V, int compact, int width,
                                                            const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ row_map,
                                                            const int32_t* __restrict__ targets, int64_t n_sel, int P,
                                                            float* __restrict__ out, unsigned* nonfinite) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_sel * P) return;
  const int pos = idx[i];
  if (pos < 0 || pos >= width) { out[i] = 0.f; return; }
  const int64_t s = i / P;
  const int64_t trow = row_map ? (int64_t)row_map[s] : s;
  const float* row = compact ? logits + (size_t)i * V : logits + ((size_t)trow * width + pos) * V;
  float mx = row[0];
  for (int v = 1; v < V; ++v) mx = fmaxf(mx, row[v]);
  float sum = 0.f;
  for (int v = 0; v < V; ++v) sum += expf(row[v] - mx);
  const float lp = row[targets[i]] - mx - logf(sum);
  if (nonfinite && !(fabsf(mx) <= 3.0e38f)) *nonfinite = 1u;      // NaN or inf anywhere in the row ends up in mx or in lp
  if (nonfinite && lp != lp) *nonfinite = 1u;
  out[i] = lp;
}

int launch_logprob_gather(hipStream_t st, const float* logits, int V, int compact, int width, const int32_t* idx,
                          const int32_t* row_map, const int32_t* targets, int64_t n_sel, int P, float* out, unsigned* nonfinite) {
  const int64_t n = n_sel * P;
  if (n == 0) return 0;
  hipLaunchKernelGGL(logprob_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, logits, V, compact, width, idx,
                     row_map, targets, n_sel, P, out, nonfinite);
  PG_HIP(hipGetLastError());
  return 0;
}

__global__ void iter_counter_kernel(int32_t* d_iter, int set, int value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *d_iter = set ? value : *d_iter + 1;
}
int launch_iter_counter(hipStream_t st, int32_t* d_iter, bool set, int value) {
  hipLaunchKernelGGL(iter_counter_kernel, dim3(1), dim3(64), 0, st, d_iter, set ? 1 : 0, value);
  PG_HIP(hipGetLastError());
  return 0;
}

static SampleArgs make_sample_args(const pg_sample_params* p, int iteration) {
  SampleArgs a;
  a.top_k = p->top_k;
  a.sample = iteration < p->burnin ? 1 : 0;   // sample=(ii < burnin), esm_sampler.py:231
  a.use_temp = (p->temperature == p->temperature) ? 1 : 0;  // NaN == None
  a.temperature = a.use_temp ? p->temperature : 1.0f;
  a.n_valid = p->n_valid;
  for (int j = 0; j < 32; ++j) a.valid_idx[j] = j < p->n_valid ? p->valid_idx[j] : 0;
  a.seed_lo = (uint32_t)(p->rng_seed & 0xffffffffu);
  a.seed_hi = (uint32_t)(p->rng_seed >> 32);
  a.stream = p->rng_stream;
  a.row_id_base = p->row_id_base;
  a.iter = (uint32_t)(p->iter_base + iteration);
  a.burnin = p->burnin;
  return a;
}

// graph state block: iteration counter + the call's sampling parameters (passed by value: no host buffer has to outlive the call)
__global__ void graph_state_kernel(int32_t* d_iter, int value, SampleArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *d_iter = value;
    *(SampleArgs*)(d_iter + kGraphArgsOffset) = a;
  }
}
size_t graph_state_bytes() { return kGraphArgsOffset * 4 + sizeof(SampleArgs); }
int launch_graph_state(hipStream_t st, int32_t* d_iter, int iteration, const pg_sample_params* p) {
  hipLaunchKernelGGL(graph_state_kernel, dim3(1), dim3(64), 0, st, d_iter, iteration, make_sample_args(p, 0));
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_mask_scatter(hipStream_t st, int32_t* tokens, int width, const int32_t* idx, const int32_t* row_map,
                        int64_t n_sel, int P, int mask_idx, const int32_t* d_iter) {
  const int64_t n = n_sel * P;
  if (n == 0) return 0;
  hipLaunchKernelGGL(mask_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tokens, width, idx, row_map,
                     n_sel, P, mask_idx, d_iter);
  PG_HIP(hipGetLastError());
  return 0;
}

int launch_sample_writeback(hipStream_t st, int32_t* tokens, int width, const float* logits, int V, int compact,
                            const int32_t* idx, const int32_t* row_map, int64_t n_sel, int P, const pg_sample_params* p,
                            int iteration, int32_t* sampled_tokens, const int32_t* d_iter, unsigned* nonfinite) {
  if (p->n_valid < 1 || p->n_valid > 32) return fail(1, "sample: n_valid must be in 1..32");
  for (int j = 0; j < p->n_valid; ++j)
    if (p->valid_idx[j] < 0 || p->valid_idx[j] >= V) return fail(1, "sample: valid_idx out of range");
  const int64_t n = n_sel * P;
  if (n == 0) return 0;
  const SampleArgs a = make_sample_args(p, iteration);
  hipLaunchKernelGGL(sample_writeback_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, tokens, width, logits, V,
                     compact, idx, row_map, n_sel, P, a, sampled_tokens, d_iter, nonfinite);      // one wave per draw
  PG_HIP(hipGetLastError());
  return 0;
}

}  // namespace pg
