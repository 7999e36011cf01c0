// pg_comm_* / pg_gather_tokens: the ONE collective of a sharded Gibbs job behind the C ABI, so that a caller that
// binds include/pgibbs.h without torch has the multi-GPU tail too.  RCCL over xGMI, one process per GPU, one communicator per
// process.  librccl.so is opened at run time (dlopen): single-GPU users need no RCCL -- neither at run time nor at BUILD time (the
// five entry points and two types used are declared below, not taken from <rccl/rccl.h>) -- and inside a torch process the copy
// torch already mapped is the one that is found: one RCCL per process.
//
// What is gathered: the final token buffers the reference untokenises (/root/reference/src/pgen/esm_sampler.py:236-239).
//
// The bookkeeping of the gather (which form, block size, scratch layout, per-rank pack offsets) is GatherPlan + gather_with<Ops>:
// one code path for the real thing (Ops = HIP copies + ncclAllGather on a stream) and for the host rehearsal
// (pg_dbg_gather_tokens_host: Ops = memcpy + an all-gather injected by the caller), so that world sizes 2 ... 8 with ragged and
// empty shards are executed by the CPU suite although the builder's boxes have one GPU (tests/test_gather_rehearsal.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "pg_common.h"
#include "pgibbs.h"

namespace {
// ---- the slice of the RCCL (= NCCL) API this file uses, declared locally: stable C ABI since NCCL 2.x
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;      // enum in the headers; 0 = ncclSuccess
constexpr int kNcclSuccess = 0;
constexpr int kNcclInt32 = 2;  // ncclDataType_t: ncclInt8 0, ncclUint8 1, ncclInt32 2
static_assert(sizeof(ncclUniqueId) == PG_COMM_ID_BYTES, "pgibbs.h and the NCCL ABI disagree about the size of a communicator id");

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why = "dlopen failed";
};

// the loader's state: filled exactly once (concurrent first calls wait for each other), read-only afterwards
Rccl& rccl_state() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already mapped (torch's) first, then the loader's search path, then the ROCm install
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if ((r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.so)
      for (const char* n : names) {
        if ((r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        const char* e = dlerror();
        if (e) r.why = e;
      }
    if (!r.so) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
      r.why = "librccl.so lacks an expected symbol";
      r.so = nullptr;
    }
  });
  return r;
}
Rccl* rccl() {
  Rccl& r = rccl_state();
  return r.so ? &r : nullptr;
}

int no_rccl() { return pg::fail(PG_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: " + rccl_state().why); }

#define PG_NCCL(expr)                                                                                        \
  do {                                                                                                       \
    ncclResult_t _r = (expr);                                                                                \
    if (_r != kNcclSuccess) return pg::fail(PG_ERR_HIP, std::string(#expr) + ": " + R->GetErrorString(_r));  \
  } while (0)

// restores the caller's current device on every exit path
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() { (void)hipGetDevice(&prev); }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ---- the gather's bookkeeping, independent of where the bytes live ---------------------------------------------------------
struct GatherPlan {
  bool equal = true;          // every rank contributes `rows` rows: all-gather straight into the output
  bool empty = false;         // nothing to move (every count is zero)
  int64_t mx = 0;             // rows of a padded block (ragged form)
  size_t row_bytes = 0, block_bytes = 0;
  size_t scratch_bytes = 0;   // ragged form: [send block | world recv blocks]
  std::vector<size_t> out_off, live_bytes;   // per rank: byte offset of its rows in the output, bytes of its live rows
};

int make_plan(int rank, int world, int64_t rows, int width, const int64_t* counts, bool force_padded, GatherPlan* p) {
  if (world < 1 || rank < 0 || rank >= world) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: rank outside [0, world)");
  if (rows < 0 || width < 1) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: bad shape");
  p->row_bytes = (size_t)width * 4;
  p->mx = rows;
  p->equal = true;
  p->out_off.assign(world, 0);
  p->live_bytes.assign(world, (size_t)rows * p->row_bytes);
  if (counts) {
    if (counts[rank] != rows) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: counts[rank] differs from rows");
    for (int r = 0; r < world; ++r) {
      if (counts[r] < 0) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: negative count");
      p->mx = counts[r] > p->mx ? counts[r] : p->mx;
      p->equal &= counts[r] == rows;
      p->live_bytes[r] = (size_t)counts[r] * p->row_bytes;
    }
    if (force_padded) p->equal = false;
  }
  size_t off = 0;
  for (int r = 0; r < world; ++r) {
    p->out_off[r] = off;
    off += p->live_bytes[r];
  }
  p->empty = p->equal ? rows == 0 : p->mx == 0;
  p->block_bytes = (size_t)p->mx * p->row_bytes;
  p->scratch_bytes = p->equal ? 0 : p->block_bytes * (size_t)(world + 1);
  return PG_OK;
}

// Ops: int zero(void*, size_t); int copy(void* dst, const void* src, size_t); int allgather(const void* send, void* recv, size_t n_int32);
//      int scratch(size_t bytes, char** out)
template <typename Ops>
int gather_with(Ops& ops, const GatherPlan& p, int rank, int world, const int32_t* local, int64_t rows, int width, int32_t* out) {
  int rc;
  if (p.empty) return PG_OK;
  if (p.equal)                                    // equal shards (256 chains over 1/2/4/8 GPUs): straight into the output
    return ops.allgather(local, out, (size_t)rows * width);
  // ragged shards: every rank contributes a block of max(counts) rows (its own rows first, zero-filled behind them), then the
  // live rows of each block are packed into the output in rank order
  char* send = nullptr;
  if ((rc = ops.scratch(p.scratch_bytes, &send))) return rc;
  char* recv = send + p.block_bytes;
  if ((rc = ops.zero(send, p.block_bytes))) return rc;
  if (rows > 0 && (rc = ops.copy(send, local, (size_t)rows * p.row_bytes))) return rc;
  if ((rc = ops.allgather(send, recv, (size_t)p.mx * width))) return rc;
  for (int r = 0; r < world; ++r)
    if (p.live_bytes[r] && (rc = ops.copy((char*)out + p.out_off[r], recv + (size_t)r * p.block_bytes, p.live_bytes[r]))) return rc;
  return PG_OK;
}
}  // namespace

struct pg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  void* scratch = nullptr;       // padded blocks of a ragged gather
  size_t scratch_bytes = 0;
};

namespace {
struct DeviceOps {
  Rccl* R;
  pg_comm* c;
  hipStream_t s;
  int zero(void* p, size_t n) { PG_HIP(hipMemsetAsync(p, 0, n, s)); return PG_OK; }
  int copy(void* d, const void* src, size_t n) { PG_HIP(hipMemcpyAsync(d, src, n, hipMemcpyDeviceToDevice, s)); return PG_OK; }
  int allgather(const void* send, void* recv, size_t n) { PG_NCCL(R->AllGather(send, recv, n, kNcclInt32, c->comm, s)); return PG_OK; }
  int scratch(size_t need, char** out) {
    if (c->scratch_bytes < need) {
      if (c->scratch) {
        PG_HIP(hipStreamSynchronize(s));          // an earlier gather on this stream may still be reading it
        PG_HIP(hipFree(c->scratch));
      }
      c->scratch = nullptr, c->scratch_bytes = 0;
      PG_HIP(hipMalloc(&c->scratch, need));
      c->scratch_bytes = need;
    }
    *out = (char*)c->scratch;
    return PG_OK;
  }
};

struct HostOps {
  pg_allgather_fn ag;
  void* ctx;
  std::vector<char> buf;
  int zero(void* p, size_t n) { memset(p, 0, n); return PG_OK; }
  int copy(void* d, const void* src, size_t n) { memcpy(d, src, n); return PG_OK; }
  int allgather(const void* send, void* recv, size_t n) {
    return ag(ctx, send, recv, n) == 0 ? PG_OK : pg::fail(PG_ERR_HIP, "pg_dbg_gather_tokens_host: the injected all-gather failed");
  }
  int scratch(size_t need, char** out) {
    buf.assign(need, (char)0x5a);                 // poison: a byte of padding that reaches the output shows up in the tests
    *out = buf.data();
    return PG_OK;
  }
};
}  // namespace

extern "C" {

int pg_comm_unique_id(void* id_out) {
  if (!id_out) return pg::fail(PG_ERR_INVALID, "pg_comm_unique_id: null argument");
  Rccl* R = rccl();
  if (!R) return no_rccl();
  ncclUniqueId id;
  PG_NCCL(R->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return PG_OK;
}

int pg_comm_create(int rank, int world, const void* unique_id, int device_ordinal, pg_comm** out) {
  if (!out || !unique_id) return pg::fail(PG_ERR_INVALID, "pg_comm_create: null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return pg::fail(PG_ERR_INVALID, "pg_comm_create: rank outside [0, world)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return pg::fail(PG_ERR_NO_DEVICE, "pg_comm_create: no HIP device");
  if (device_ordinal < 0 || device_ordinal >= n_dev) return pg::fail(PG_ERR_INVALID, "pg_comm_create: no such device");
  Rccl* R = rccl();
  if (!R) return no_rccl();
  DeviceGuard guard;                                  // the caller's device comes back on every path, also when hipSetDevice fails
  PG_HIP(hipSetDevice(device_ordinal));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  pg_comm* c = new pg_comm;
  c->rank = rank, c->world = world, c->device = device_ordinal;
  ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (r != kNcclSuccess) {
    delete c;
    return pg::fail(PG_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
  }
  *out = c;
  return PG_OK;
}

void pg_comm_destroy(pg_comm* c) {
  if (!c) return;
  Rccl* R = rccl();
  DeviceGuard guard;
  (void)hipSetDevice(c->device);
  if (c->scratch) (void)hipFree(c->scratch);
  if (R && c->comm) (void)R->CommDestroy(c->comm);
  delete c;
}

int pg_comm_rank(const pg_comm* c) { return c ? c->rank : -1; }
int pg_comm_world(const pg_comm* c) { return c ? c->world : 0; }

int pg_gather_tokens(pg_comm* c, void* hip_stream, const int32_t* d_local, int64_t rows, int width, const int64_t* counts,
                     int32_t* d_out) {
  if (!c || !d_out) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: null argument");
  if (rows > 0 && !d_local) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: null local buffer");
  GatherPlan plan;
  // PGIBBS_GATHER_FORCE_PADDED: tests run the ragged form on equal shards (one GPU per box)
  int rc = make_plan(c->rank, c->world, rows, width, counts, counts && getenv("PGIBBS_GATHER_FORCE_PADDED"), &plan);
  if (rc) return rc;
  Rccl* R = rccl();
  if (!R) return no_rccl();
  DeviceGuard guard;
  PG_HIP(hipSetDevice(c->device));
  DeviceOps ops{R, c, (hipStream_t)hip_stream};
  return gather_with(ops, plan, c->rank, c->world, d_local, rows, width, d_out);
}

int pg_dbg_gather_tokens_host(int rank, int world, const int32_t* local, int64_t rows, int width, const int64_t* counts,
                              int force_padded, pg_allgather_fn allgather, void* ctx, int32_t* out) {
  if (!allgather || !out) return pg::fail(PG_ERR_INVALID, "pg_dbg_gather_tokens_host: null argument");
  if (rows > 0 && !local) return pg::fail(PG_ERR_INVALID, "pg_dbg_gather_tokens_host: null local buffer");
  GatherPlan plan;
  int rc = make_plan(rank, world, rows, width, counts, counts && force_padded, &plan);
  if (rc) return rc;
  HostOps ops{allgather, ctx, {}};
  return gather_with(ops, plan, rank, world, local, rows, width, out);
}

int pg_dbg_gather_plan(int rank, int world, int64_t rows, int width, const int64_t* counts, int force_padded, int64_t* out7,
                       int64_t* out_off_bytes) {
  if (!out7) return pg::fail(PG_ERR_INVALID, "pg_dbg_gather_plan: null argument");
  GatherPlan p;
  int rc = make_plan(rank, world, rows, width, counts, counts && force_padded, &p);
  if (rc) return rc;
  out7[0] = p.equal, out7[1] = p.empty, out7[2] = p.mx, out7[3] = (int64_t)p.row_bytes, out7[4] = (int64_t)p.block_bytes;
  out7[5] = (int64_t)p.scratch_bytes, out7[6] = (int64_t)(p.out_off[world - 1] + p.live_bytes[world - 1]);
  if (out_off_bytes)
    for (int r = 0; r < world; ++r) out_off_bytes[r] = (int64_t)p.out_off[r];
  return PG_OK;
}

}  // extern "C"
