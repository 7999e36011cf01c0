// pg_comm_* / pg_gather_tokens: the ONE collective of a sharded Gibbs job behind the C ABI, so that a caller that
// binds include/pgibbs.h without torch has the multi-GPU tail too.  RCCL over xGMI, one process per GPU, one communicator per
// process.  librccl.so is opened at run time (dlopen): single-GPU users need no RCCL, and inside a torch process the copy torch
// already mapped is the one that is found -- one RCCL per process.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pg_common.h"
#include "pgibbs.h"

static_assert(PG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pgibbs.h and rccl.h disagree about the size of a communicator id");

namespace {
struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::string g_why = "dlopen failed";

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.so ? &r : nullptr;
  tried = true;
  // a copy that is already mapped (torch's) first, then the loader's search path, then the ROCm install
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names)
    if ((r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!r.so)
    for (const char* n : names) {
      if ((r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      const char* e = dlerror();
      if (e) g_why = e;
    }
  if (!r.so) return nullptr;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
    g_why = "librccl.so lacks an expected symbol";
    r.so = nullptr;
    return nullptr;
  }
  return &r;
}

int no_rccl() { return pg::fail(PG_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: " + g_why); }

#define PG_NCCL(expr)                                                                                        \
  do {                                                                                                       \
    ncclResult_t _r = (expr);                                                                                \
    if (_r != ncclSuccess) return pg::fail(PG_ERR_HIP, std::string(#expr) + ": " + R->GetErrorString(_r));   \
  } while (0)
}  // namespace

struct pg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  void* scratch = nullptr;       // padded blocks of a ragged gather
  size_t scratch_bytes = 0;
};

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
  int prev = -1;
  (void)hipGetDevice(&prev);
  PG_HIP(hipSetDevice(device_ordinal));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  pg_comm* c = new pg_comm;
  c->rank = rank, c->world = world, c->device = device_ordinal;
  ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
  if (prev >= 0) (void)hipSetDevice(prev);
  if (r != ncclSuccess) {
    delete c;
    return pg::fail(PG_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
  }
  *out = c;
  return PG_OK;
}

void pg_comm_destroy(pg_comm* c) {
  if (!c) return;
  Rccl* R = rccl();
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  if (c->scratch) (void)hipFree(c->scratch);
  if (R && c->comm) (void)R->CommDestroy(c->comm);
  if (prev >= 0) (void)hipSetDevice(prev);
  delete c;
}

int pg_comm_rank(const pg_comm* c) { return c ? c->rank : -1; }
int pg_comm_world(const pg_comm* c) { return c ? c->world : 0; }

int pg_gather_tokens(pg_comm* c, void* hip_stream, const int32_t* d_local, int64_t rows, int width, const int64_t* counts,
                     int32_t* d_out) {
  if (!c || !d_out) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: null argument");
  if (rows < 0 || width < 1) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: bad shape");
  if (rows > 0 && !d_local) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: null local buffer");
  Rccl* R = rccl();
  if (!R) return no_rccl();
  hipStream_t s = (hipStream_t)hip_stream;
  int64_t mx = rows;
  bool equal = true;
  if (counts) {
    if (counts[c->rank] != rows) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: counts[rank] differs from rows");
    for (int r = 0; r < c->world; ++r) {
      if (counts[r] < 0) return pg::fail(PG_ERR_INVALID, "pg_gather_tokens: negative count");
      mx = counts[r] > mx ? counts[r] : mx;
      equal &= counts[r] == rows;
    }
  }
  if (counts && getenv("PGIBBS_GATHER_FORCE_PADDED")) equal = false;      // tests: the ragged form on equal shards (one GPU per box)
  int prev = -1;
  (void)hipGetDevice(&prev);
  struct Restore {
    int prev;
    ~Restore() {
      if (prev >= 0) (void)hipSetDevice(prev);
    }
  } restore{prev};
  PG_HIP(hipSetDevice(c->device));
  const size_t row_bytes = (size_t)width * 4;
  if (equal) {                                    // equal shards (256 chains over 1/2/4/8 GPUs): straight into the output
    if (rows == 0) return PG_OK;
    PG_NCCL(R->AllGather(d_local, d_out, (size_t)rows * width, ncclInt32, c->comm, s));
    return PG_OK;
  }
  // ragged shards: every rank contributes a block of max(counts) rows (its own rows first), then the live rows of each
  // block are packed into the output in rank order
  if (mx == 0) return PG_OK;
  const size_t block = (size_t)mx * row_bytes, need = block * (size_t)(c->world + 1);
  if (c->scratch_bytes < need) {
    if (c->scratch) PG_HIP(hipFree(c->scratch));
    c->scratch = nullptr, c->scratch_bytes = 0;
    PG_HIP(hipMalloc(&c->scratch, need));
    c->scratch_bytes = need;
  }
  char* send = (char*)c->scratch;
  char* recv = send + block;
  PG_HIP(hipMemsetAsync(send, 0, block, s));
  if (rows > 0) PG_HIP(hipMemcpyAsync(send, d_local, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, s));
  PG_NCCL(R->AllGather(send, recv, (size_t)mx * width, ncclInt32, c->comm, s));
  size_t off = 0;
  for (int r = 0; r < c->world; ++r) {
    const size_t nb = (size_t)counts[r] * row_bytes;
    if (nb) PG_HIP(hipMemcpyAsync((char*)d_out + off, recv + (size_t)r * block, nb, hipMemcpyDeviceToDevice, s));
    off += nb;
  }
  return PG_OK;
}

}  // extern "C"
