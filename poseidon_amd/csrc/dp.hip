// Data-parallel gradient exchange behind the C ABI (include/scot_hip.h: scot_dp_*).
//
// The reference reaches its one collective per step — the mean all-reduce of every gradient — through torch DDP (HF Trainer / accelerate,
// scOT/train.py:281 per-device batch, README.md:50-57 the accelerate launch).  Here the gradients are ranges of ONE fp32 arena, so the
// exchange of a range is one in-place RCCL all-reduce on a communication stream; nothing else of DDP (buckets, hooks, copies) is needed.
// A host that has torch.distributed takes poseidon_amd/dp.py with backend="torch" (the default); these entry points serve a host
// WITHOUT it (a C / C++ caller above the C ABI) and are what dp.py's backend="native" calls.
//
// RCCL is resolved at run time: the library that is already mapped into the process (torch's librccl.so.1) is preferred, then the
// loader path, then /opt/rocm/lib — libscot_hip.so itself has no link-time dependency on it, so every other entry point keeps working
// on a box without RCCL and scot_dp_init fails loudly (SCOT_ERR_UNSUPPORTED + a line on stderr) there.
// One communicator per process (one process per GPU); not thread-safe against itself, like the rest of the host-side ABI.
#include "common.h"
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

namespace {

struct RcclUniqueId { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
typedef void* RcclComm;
enum { RCCL_SUM = 0, RCCL_F16 = 6, RCCL_F32 = 7, RCCL_BF16 = 9 };   // ncclRedOp_t / ncclDataType_t values of rccl.h

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclUniqueId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;
RcclComm g_comm = nullptr;
int g_world = 0, g_rank = -1;

bool rccl_load() {
  if (g_rccl.handle) return true;
  void* h = nullptr;
  const char* mapped[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : mapped)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already uses (torch's), if any
  const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : paths)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    fprintf(stderr, "[scot_dp] RCCL not found (librccl.so.1): %s\n", dlerror());
    return false;
  }
  Rccl r;
  r.handle = h;
  r.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (int (*)(RcclComm))dlsym(h, "ncclCommDestroy");
  r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  r.GetVersion = (int (*)(int*))dlsym(h, "ncclGetVersion");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) {
    fprintf(stderr, "[scot_dp] the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy\n");
    dlclose(h);
    return false;
  }
  g_rccl = r;
  return true;
}

int rccl_fail(const char* what, int rc) {
  fprintf(stderr, "[scot_dp] %s failed: %s (%d)\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", rc);
  return SCOT_ERR_LAUNCH;
}

}  // namespace

// include/scot_hip.h: rank 0 draws the 128-byte rendezvous token (ncclGetUniqueId); the host hands it to every rank by its own means
// (a file, MPI, torch's store — poseidon_amd/dp.py broadcasts it through the process group it already has).
extern "C" int scot_dp_unique_id(void* id128) {
  if (!id128) return SCOT_ERR_SHAPE;
  if (!rccl_load()) return SCOT_ERR_UNSUPPORTED;
  RcclUniqueId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return SCOT_OK;
}

// Joins the communicator of `world` ranks as `rank` on the CURRENT HIP device (one process per GPU: hipSetDevice(LOCAL_RANK) first).
// Collective over all ranks; blocks until every rank has called it.
extern "C" int scot_dp_init(const void* id128, int rank, int world) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) return SCOT_ERR_SHAPE;
  if (g_comm) return SCOT_ERR_UNSUPPORTED;                  // one communicator per process: scot_dp_finalize first
  if (!rccl_load()) return SCOT_ERR_UNSUPPORTED;
  RcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  RcclComm comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
  if (rc || !comm) return rccl_fail("ncclCommInitRank", rc);
  g_comm = comm; g_world = world; g_rank = rank;
  return SCOT_OK;
}

// In-place SUM over ranks of grads[0..n) (dtype 0: fp32; 1: the bfloat16 wire format of scot_dp_pack — bfloat16 in BOTH builds of the
// library), enqueued on comm_stream.  The mean's 1/world is the caller's (scot_dp_pack folds it into the wire format; the fp32 path
// scales the range before the call, as dp.py does): the sum of pre-scaled terms is what keeps a 16-bit wire from overflowing.
extern "C" int scot_dp_allreduce_bucket(void* grads, size_t n, int dtype, hipStream_t comm_stream) {
  if (!g_comm) return SCOT_ERR_UNSUPPORTED;
  if (dtype != SCOT_F32 && dtype != SCOT_BF16) return SCOT_ERR_DTYPE;
  if (n == 0) return SCOT_OK;
  if (!grads) return SCOT_ERR_SHAPE;
  const int rc = g_rccl.AllReduce(grads, grads, n, dtype == SCOT_F32 ? RCCL_F32 : RCCL_BF16, RCCL_SUM, g_comm, comm_stream);
  return rc ? rccl_fail("ncclAllReduce", rc) : SCOT_OK;
}

// 0 before scot_dp_init / after scot_dp_finalize, else the communicator's size; scot_dp_rank: -1 / this process's rank.
extern "C" int scot_dp_world(void) { return g_comm ? g_world : 0; }
extern "C" int scot_dp_rank(void) { return g_comm ? g_rank : -1; }

// Destroys the communicator (the caller has synchronised the streams its collectives ran on).  Idempotent.
extern "C" int scot_dp_finalize(void) {
  if (!g_comm) return SCOT_OK;
  const int rc = g_rccl.CommDestroy(g_comm);
  g_comm = nullptr; g_world = 0; g_rank = -1;
  return rc ? rccl_fail("ncclCommDestroy", rc) : SCOT_OK;
}
