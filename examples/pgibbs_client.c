/* A caller of the C ABI (include/pgibbs.h) that is not Python: plain C99, no torch, no HIP headers.
 *
 * What a maintainer binding the engine from another language would write: build a model from named fp32
 * tensors, run the whole Gibbs loop of ESM_sampler.generate (/root/reference/src/pgen/esm_sampler.py:209-234) in one call, then
 * the job's one collective (a world of one here) on a device buffer.  tests/test_gpu_c_client.py compiles this file with gcc,
 * runs it on the MI355X and checks its tokens and emitted logits bit for bit against the Python path on the same weights.
 *
 *   gcc -std=c99 -O1 -I include examples/pgibbs_client.c -L protein_gibbs_sampler_amd/lib -lpgibbs -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/protein_gibbs_sampler_amd/lib -Wl,-rpath,/opt/rocm/lib -o build/pgibbs_client
 *   build/pgibbs_client weights.bin out.bin
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pgibbs.h"

/* the four HIP runtime calls the collective part needs (device memory belongs to the caller there) */
extern int hipMalloc(void** p, size_t n);
extern int hipFree(void* p);
extern int hipMemcpy(void* dst, const void* src, size_t n, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

enum { D = 128, LAYERS = 2, HEADS = 2, FFN = 256, MAXPOS = 64, VOCAB = 33, MAXT = 512 };

static pg_tensor tensors[MAXT];
static char names[MAXT][96];
static int n_tensors = 0;
static uint32_t lcg = 20260930u;

static float* add(const char* name, int64_t numel, float scale, float offset) {
  float* v = (float*)malloc((size_t)numel * sizeof(float));
  for (int64_t i = 0; i < numel; ++i) {
    lcg = lcg * 1664525u + 1013904223u;
    v[i] = offset + scale * ((float)(lcg >> 8) * (1.0f / 8388608.0f) - 1.0f);
  }
  snprintf(names[n_tensors], sizeof names[0], "%s", name);
  tensors[n_tensors].name = names[n_tensors];
  tensors[n_tensors].data = v;
  tensors[n_tensors].numel = numel;
  ++n_tensors;
  return v;
}
static void linear(const char* prefix, int out, int in) {
  char n[96];
  snprintf(n, sizeof n, "%s.weight", prefix); add(n, (int64_t)out * in, 0.08f, 0.f);
  snprintf(n, sizeof n, "%s.bias", prefix);   add(n, out, 0.05f, 0.f);
}
static void layer_norm(const char* prefix) {
  char n[96];
  snprintf(n, sizeof n, "%s.weight", prefix); add(n, D, 0.1f, 1.f);
  snprintf(n, sizeof n, "%s.bias", prefix);   add(n, D, 0.1f, 0.f);
}
#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != PG_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pg_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s weights_out.bin results_out.bin\n", argv[0]); return 2; }
  if (pg_device_count() < 1) { fprintf(stderr, "no MI355X visible\n"); return 3; }

  /* ---- the fair-esm ESM-1b state dict, by name ---- */
  add("embed_tokens.weight", (int64_t)VOCAB * D, 0.5f, 0.f);
  add("embed_positions.weight", (int64_t)(MAXPOS + 1 + 1) * D, 0.5f, 0.f);
  layer_norm("emb_layer_norm_before");
  for (int l = 0; l < LAYERS; ++l) {
    char p[64], n[96];
    snprintf(p, sizeof p, "layers.%d", l);
    const char* proj[4] = {"q_proj", "k_proj", "v_proj", "out_proj"};
    for (int j = 0; j < 4; ++j) { snprintf(n, sizeof n, "%s.self_attn.%s", p, proj[j]); linear(n, D, D); }
    snprintf(n, sizeof n, "%s.self_attn_layer_norm", p); layer_norm(n);
    snprintf(n, sizeof n, "%s.fc1", p); linear(n, FFN, D);
    snprintf(n, sizeof n, "%s.fc2", p); linear(n, D, FFN);
    snprintf(n, sizeof n, "%s.final_layer_norm", p); layer_norm(n);
  }
  layer_norm("emb_layer_norm_after");
  linear("lm_head.dense", D, D);
  layer_norm("lm_head.layer_norm");
  add("lm_head.bias", VOCAB, 0.05f, 0.f);

  FILE* f = fopen(argv[1], "wb");
  if (!f) return 4;
  fwrite(&n_tensors, sizeof(int), 1, f);
  for (int i = 0; i < n_tensors; ++i) {
    int32_t len = (int32_t)strlen(names[i]);
    fwrite(&len, 4, 1, f);
    fwrite(names[i], 1, (size_t)len, f);
    fwrite(&tensors[i].numel, 8, 1, f);
    fwrite(tensors[i].data, 4, (size_t)tensors[i].numel, f);
  }
  fclose(f);

  pg_model_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.arch = PG_ARCH_ESM1B; cfg.vocab = VOCAB; cfg.d_model = D; cfg.n_layers = LAYERS; cfg.n_heads = HEADS; cfg.d_ffn = FFN;
  cfg.max_positions = MAXPOS; cfg.pad_idx = 1; cfg.mask_idx = 32; cfg.cls_idx = 0; cfg.eos_idx = 2; cfg.token_dropout = 1;
  cfg.max_msa_rows = 0; cfg.layer_norm_eps = 1e-5f;
  pg_engine* eng = NULL;
  CHECK(pg_engine_create(&cfg, tensors, n_tensors, 0, PG_PREC_BF16, &eng));

  /* ---- 3 chains of 18 residues (T = 20), 2 iterations x 4 positions, positions from the CPython-exact generator ---- */
  enum { B = 3, T = 20, ITERS = 2, P = 4 };
  int32_t tokens[B * T], start[B * T], idx[ITERS * B * P], population[T - 2], sampled_tokens[ITERS * B * P];
  static float sampled_logits[ITERS * B * P * VOCAB];
  for (int b = 0; b < B; ++b) {
    tokens[b * T] = 0;
    for (int t = 1; t < T - 1; ++t) tokens[b * T + t] = 4 + (7 * b + 3 * t) % 20;
    tokens[b * T + T - 1] = 2;
  }
  memcpy(start, tokens, sizeof tokens);
  for (int t = 0; t < T - 2; ++t) population[t] = t + 1;
  pg_pyrandom* rng = pg_pyrandom_create();
  uint32_t seed_words[1] = {12345u};
  CHECK(pg_pyrandom_seed(rng, seed_words, 1));
  CHECK(pg_pyrandom_sample_table(rng, population, T - 2, P, (int64_t)ITERS * B, idx));       /* random.sample, row by row */
  pg_pyrandom_destroy(rng);

  pg_sample_params sp;
  memset(&sp, 0, sizeof sp);
  sp.mask = 1; sp.mask_idx = 32; sp.top_k = 0; sp.burnin = INT32_MAX; sp.temperature = 1.0f; sp.n_valid = 20;
  for (int j = 0; j < 20; ++j) sp.valid_idx[j] = 4 + j;
  sp.rng_seed = 99; sp.rng_stream = 0; sp.row_id_base = 0; sp.iter_base = 0;
  CHECK(pg_esm_gibbs_run(eng, tokens, B, T, idx, ITERS, P, &sp, sampled_logits, sampled_tokens));

  /* ---- the job's one collective, world of one: the library opens RCCL itself ---- */
  unsigned char id[PG_COMM_ID_BYTES];
  pg_comm* comm = NULL;
  int32_t gathered[B * T];
  void *d_in = NULL, *d_out = NULL;
  int64_t counts[1] = {B};
  CHECK(pg_comm_unique_id(id));
  CHECK(pg_comm_create(0, 1, id, 0, &comm));
  if (hipMalloc(&d_in, sizeof tokens) || hipMalloc(&d_out, sizeof tokens) || hipMemcpy(d_in, tokens, sizeof tokens, 1)) return 5;
  CHECK(pg_gather_tokens(comm, NULL, (const int32_t*)d_in, B, T, counts, (int32_t*)d_out));
  if (hipDeviceSynchronize() || hipMemcpy(gathered, d_out, sizeof tokens, 2)) return 6;
  pg_comm_destroy(comm);
  hipFree(d_in);
  hipFree(d_out);
  if (memcmp(gathered, tokens, sizeof tokens)) { fprintf(stderr, "gather returned other tokens\n"); return 7; }

  f = fopen(argv[2], "wb");
  if (!f) return 4;
  fwrite(start, 4, B * T, f);
  fwrite(idx, 4, ITERS * B * P, f);
  fwrite(tokens, 4, B * T, f);
  fwrite(sampled_tokens, 4, ITERS * B * P, f);
  fwrite(sampled_logits, 4, ITERS * B * P * VOCAB, f);
  fclose(f);
  pg_engine_destroy(eng);
  printf("%s: %d tensors, %d draws, first chain now", pg_version(), n_tensors, ITERS * B * P);
  for (int t = 0; t < T; ++t) printf(" %d", tokens[t]);
  printf("\n");
  return 0;
}
