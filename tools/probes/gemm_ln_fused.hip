// PROBE / RECORD, not part of libpgibbs.so (moved out of csrc/ in round 4).  LayerNorm inside the residual GEMM
// (EPI_F32_RESID_LN, round 3, commit aee158c): bit-identical with the stand-alone LayerNorm kernel, measured SLOWER on every
// configuration (ESM-1b config 2: 91.5 vs 87.8 ms per iteration; ESM-MSA-1b config 4: 153.7 vs 150.4; A/B table in
// profiles/r03_layernorm_fusion_ab.txt, analysis in DESIGN.md section 4 "Round 3").  It lived behind PGIBBS_LN_FUSE=1 as a
// default-off path through the hot kernels; VERDICT r03 asked for it to leave the product as gemm_w4 did.  This file keeps the
// device code, the host-side fusion condition, the engine hook and the debug entry + test so the experiment can be re-applied:
//   git show aee158c   (the in-tree form: gemm_epilogue.h, gemm_bf16.hip, engine.hip, api.hip, tests/test_gpu_ln_fused.py)
// Nothing here is compiled by csrc/Makefile.
#if 0
// ---- gemm_epilogue.h: operands + the arrival counter / normalising workgroup ----
// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm inside the residual GEMM (EPI_F32_RESID_LN).  x += ctx W^T + b is followed, in the forward pass, by h = LN(x): a
// kernel that re-reads from HBM the 338 MB the GEMM has just written (config 2: 85 us, 66 times per iteration, at the HBM
// roofline -- pure avoidable traffic).  A row can only be normalised when all of its column tiles are done, so every workgroup,
// after storing its tile, counts itself into a per-row-panel counter; the LAST one to arrive for a panel normalises that
// panel's rows -- one wave per row, the very code of the stand-alone kernel (ln_row.h), so h is bit-identical with the unfused
// path and results do not depend on which workgroup happens to be last, nor on whether a launch fuses at all.
// Coherence without cache maintenance: the launcher fuses only when all column tiles of a panel run on ONE XCD (the grouped
// tile order puts them there; checked on the host, else the LayerNorm kernel follows as before).  The XCD's CUs share its L2,
// which is therefore the coherence point: a writer's stores are acknowledged by L2 (s_waitcnt vmcnt(0)) before it counts
// itself in, the counter is an L2 atomic, and the normalising workgroup reads the rows with sc0 loads (past its own L1).  An
// agent-scope release / acquire pair instead (buffer_wbl2 / buffer_inv sc1 per tile) writes back and invalidates the whole L2
// under the other CUs' main loops: measured 121 vs 90 ms per iteration.
// Counters reset themselves (the last arriver stores 0), so a zero-filled array serves every launch.
// ---------------------------------------------------------------------------------------------------------------------
struct EpiAux {
  bf16_t* h;               // LayerNorm output rows [M][d] bf16 (the next GEMM's operand)
  const float* gamma;      // the FOLLOWING LayerNorm's weight / bias [d]
  const float* beta;
  int* counters;           // one per 256-row panel of 256 x 256 tiles, then one per 64-row block of tail tiles; zero between launches
  float eps;
  int flags;               // experiments (PGIBBS_LN_FLAGS): 1 = the residual rows leave with ordinary (not streaming) stores
};

// Called by every thread of a workgroup after its tile's stores: counts the workgroup into counters[slot]; the workgroup that
// completes the n_tiles of the row block normalises its n_rows rows (x_rows = first row of the block, row length d = ldo).
template <int NW>
__device__ __forceinline__ void ln_when_panel_complete(const EpiAux& aux, int slot, int n_tiles, const float* x_rows, int64_t row0,
                                                       int n_rows, int d, int* flag /* LDS word */) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my stores of the residual tile are in the XCD's L2
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(aux.counters + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == n_tiles - 1;
    if (last) __hip_atomic_store(aux.counters + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ln_rows_l2<4>(x_rows, d, aux.h + (size_t)row0 * d, n_rows, wave, NW, d, aux.eps, aux.gamma, aux.beta, lane);
}

// ---- ln_row.h: rows read past the CU's L1 (sc0) from the XCD's L2 ----
// rows r = first, first + step, ... < n_rows of a row block: x fp32 [.][ldx] -> LayerNorm -> h bf16 [.][d] (3 d with split3);
// one wave per row, RB rows in flight per wave (the loop is latency-bound: 256 rows on 8 waves are 32 round trips to L2 / HBM
// one row at a time).  The rows were written by OTHER CUs of this XCD a moment ago (gemm_epilogue.h): they are read with
// sc0 = 1, i.e. past this CU's vector L1 from the L2 the XCD shares.
template <int RB>
__device__ __forceinline__ void ln_rows_l2(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ h, int n_rows, int first,
                                           int step, int d, float eps, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, int lane, bool split3 = false) {
  typedef float pg_f32x4_t __attribute__((ext_vector_type(4)));
  const int nch4 = d >> 2;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7fffffff, 0x00020000);
  for (int r0 = first; r0 < n_rows; r0 += step * RB) {
    float4 v[RB][kMaxCh];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int r = r0 + u * step;
#pragma unroll
      for (int i = 0; i < kMaxCh; ++i)
        if (r < n_rows && lane + 64 * i < nch4) {
          const pg_f32x4_t q = __builtin_bit_cast(pg_f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((int64_t)r * ldx * 4) + (lane + 64 * i) * 16, 0, 1 /* sc0 */));
          v[u][i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int r = r0 + u * step;
      if (r < n_rows) {
        ln_inplace(v[u], nch4, lane, d, eps, gamma, beta);
        store_row_bf16(h + (size_t)r * d * (split3 ? 3 : 1), v[u], nch4, lane, split3);
      }
    }
  }
}

// ---- gemm_bf16.hip: host-side condition (every row panel's column tiles on one XCD) ----
// EPI_F32_RESID_LN needs every row panel's column tiles on one XCD (gemm_epilogue.h): XCD x owns the contiguous tile range
// [x q + min(x, r), ...) of the grouped order, a group being gm m-panels x all n-tiles -- so every range must begin on a group
// boundary; tail row blocks must divide by 8.
bool gemm_big_can_fuse_ln(int M, int N, int K) {
  if (M % 256 || N % 256 || K % 64 || K < 128 || M < 256 || N > 2048) return false;
  const BigGeom g = big_geometry(M, N, K);
  const int tiles_n = N / 256, n_tiles = g.m_main * tiles_n, gsz = g.gm * tiles_n;
  const int q = n_tiles >> 3, r = n_tiles & 7;
  for (int x = 1; x < 8; ++x)
    if ((x * q + (x < r ? x : r)) % gsz) return false;
  return ((g.tail_rows / 64) & 7) == 0;
}

// ---- gemm_bf16.hip, end of gemm_bf16_pp_kernel ----
  if (EPI == EPI_F32_RESID_LN) {     // the last of the panel's column tiles normalises its 256 rows (gemm_epilogue.h)
    __syncthreads();                 // the staging LDS is free (flag word)
    ln_when_panel_complete<8>(aux, tile_m, tiles_n, (const float*)out + (size_t)m0 * ldo, m0, 256, ldo, (int*)smem);
  }
}

// ---- gemm_epilogue.h, end of gemm_tail_tile64 ----
  if (EPI == EPI_F32_RESID_LN)     // the last of the row block's N / 64 tiles normalises its 64 rows
    ln_when_panel_complete<NW>(aux, ln_slot, ldo / 64, (const float*)out + (size_t)m0 * ldo, m0, 64, ldo, (int*)smem);
}

// ---- engine.hip: Engine::resid_gemm_ln ----
// x[M_rows][d] += a[M_rows][K] W^T + b, then h = LayerNorm(x; ln) as the next GEMM's bf16 operand.  Big batches: ONE launch --
// the residual GEMM's workgroup that completes a row panel normalises it (gemm_epilogue.h) --, otherwise the GEMM the dispatch
// picks followed by the LayerNorm kernel.  Both forms run the same per-row LayerNorm code on the same fp32 rows: identical bits,
// so the choice is purely local (no shard-invariance concern).
int Engine::resid_gemm_ln(const bf16_t* a, const DenseW& W, float* x, int M_rows, int64_t M_real, int lda, const LnW& ln, bf16_t* h,
                          float* ws, size_t ws_bytes) {
  const int d = W.N, K = W.K;
  const bool big = fuse_ln && M_rows % 256 == 0 && d % 256 == 0 && K >= 128 && K % 64 == 0 &&
                   (long)(M_rows / 256) * (d / 256) >= 128 &&      // the shapes launch_gemm_bf16 gives to the big-tile kernels
                   gemm_big_can_fuse_ln(M_rows, d, K);              // ... with every row panel's tiles on one XCD
  if (big) {
    int rc = ln_counters.ensure((size_t)(M_rows / 64 + 8) * 4, stream);
    if (rc) return rc;
    EpiAux aux{};
    aux.h = h;
    aux.gamma = ln.g;
    aux.beta = ln.b;
    aux.counters = ln_counters.as<int>();
    aux.eps = cfg.layer_norm_eps;
    static const int ln_flags = [] { const char* e = getenv("PGIBBS_LN_FLAGS"); return e ? atoi(e) : 0; }();
    aux.flags = ln_flags;
    return timed(PC_GEMM, [&] { return launch_gemm_big(stream, a, W.w, W.b, x, M_rows, d, K, lda, K, d, EPI_F32_RESID_LN, &aux); });
  }
  int rc = timed(PC_GEMM, [&] { return launch_gemm_bf16(stream, a, W.w, W.b, x, M_rows, d, K, lda, K, d, EPI_F32_RESID, ws, ws_bytes); });
  if (rc) return rc;
  return timed(PC_LN, [&] { return launch_layernorm_bf16(stream, x, ln.g, ln.b, h, M_real, d, cfg.layer_norm_eps); });
}

// ---- api.hip: debug entry (kernel-level test) ----
int pg_dbg_gemm_resid_ln(int device, const float* a, const float* w, const float* bias, float* resid_inout, const float* gamma,
                         const float* beta, float* h_fused, float* h_kernel, int M, int N, int K, float eps, int repeats) {
  if (!a || !w || !bias || !resid_inout || !gamma || !beta || !h_fused || !h_kernel) return fail(PG_ERR_INVALID, "pg_dbg_gemm_resid_ln: null argument");
  if (M < 256 || M % 256 || N % 256 || K % 64 || K < 128 || N > 2048 || repeats < 1) return fail(PG_ERR_INVALID, "pg_dbg_gemm_resid_ln: shape");
  DeviceGuard g(-1);
  int rc = dbg_device(device);
  if (rc) return rc;
  Tmp t;
  float* tmp = (float*)t.get((size_t)std::max((size_t)M * K, (size_t)N * K) * 4);
  bf16_t* ba = (bf16_t*)t.get((size_t)M * K * 2);
  bf16_t* bw = (bf16_t*)t.get((size_t)N * K * 2);
  float* db = (float*)t.get((size_t)N * 4);
  float* dg = (float*)t.get((size_t)N * 4);
  float* dbt = (float*)t.get((size_t)N * 4);
  float* dx = (float*)t.get((size_t)M * N * 4);
  float* dx0 = (float*)t.get((size_t)M * N * 4);
  bf16_t* h1 = (bf16_t*)t.get((size_t)M * N * 2);
  bf16_t* h2 = (bf16_t*)t.get((size_t)M * N * 2);
  float* hf = (float*)t.get((size_t)M * N * 4);
  int* cnt = (int*)t.get((size_t)(M / 64 + 8) * 4);           // zero-filled by Tmp::get
  if (!tmp || !ba || !bw || !db || !dg || !dbt || !dx || !dx0 || !h1 || !h2 || !hf || !cnt) return fail(PG_ERR_HIP, "hipMalloc failed");
  PG_HIP(hipMemcpy(tmp, a, (size_t)M * K * 4, hipMemcpyHostToDevice));
  if ((rc = launch_f32_to_bf16(nullptr, tmp, ba, (int64_t)M * K, 1.f))) return rc;
  PG_HIP(hipDeviceSynchronize());
  PG_HIP(hipMemcpy(tmp, w, (size_t)N * K * 4, hipMemcpyHostToDevice));
  if ((rc = launch_f32_to_bf16(nullptr, tmp, bw, (int64_t)N * K, 1.f))) return rc;
  PG_HIP(hipMemcpy(db, bias, (size_t)N * 4, hipMemcpyHostToDevice));
  PG_HIP(hipMemcpy(dg, gamma, (size_t)N * 4, hipMemcpyHostToDevice));
  PG_HIP(hipMemcpy(dbt, beta, (size_t)N * 4, hipMemcpyHostToDevice));
  PG_HIP(hipMemcpy(dx0, resid_inout, (size_t)M * N * 4, hipMemcpyHostToDevice));
  EpiAux aux{};
  aux.h = h1;
  aux.gamma = dg;
  aux.beta = dbt;
  aux.counters = cnt;
  aux.eps = eps;
  // repeated launches: the arrival counters must reset themselves, and who arrives last varies from launch to launch
  if (!gemm_big_can_fuse_ln(M, N, K)) return fail(PG_ERR_UNSUPPORTED, "pg_dbg_gemm_resid_ln: a row panel's tiles would span XCDs at this shape");
  for (int r = 0; r < repeats; ++r) {
    PG_HIP(hipMemcpyAsync(dx, dx0, (size_t)M * N * 4, hipMemcpyDeviceToDevice, nullptr));
    PG_HIP(hipMemsetAsync(h1, 0xff, (size_t)M * N * 2, nullptr));
    if ((rc = launch_gemm_big(nullptr, ba, bw, db, dx, M, N, K, K, K, N, EPI_F32_RESID_LN, &aux))) return rc;
  }
  if ((rc = launch_layernorm_bf16(nullptr, dx, dg, dbt, h2, M, N, eps))) return rc;      // the stand-alone kernel on the same rows
  if ((rc = launch_bf16_to_f32(nullptr, h1, hf, (int64_t)M * N))) return rc;
  PG_HIP(hipDeviceSynchronize());
  PG_HIP(hipMemcpy(h_fused, hf, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  if ((rc = launch_bf16_to_f32(nullptr, h2, hf, (int64_t)M * N))) return rc;
  PG_HIP(hipDeviceSynchronize());
  PG_HIP(hipMemcpy(h_kernel, hf, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  PG_HIP(hipMemcpy(resid_inout, dx, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  std::vector<int> c((size_t)M / 64 + 8);
  PG_HIP(hipMemcpy(c.data(), cnt, c.size() * 4, hipMemcpyDeviceToHost));
  for (int v : c)
    if (v != 0) return fail(PG_ERR_HIP, "pg_dbg_gemm_resid_ln: an arrival counter did not return to zero");
  return PG_OK;
}

#endif
