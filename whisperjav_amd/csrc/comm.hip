// comm.hip -- the one collective of the path behind the C ABI: broadcast of the packed weight blob over RCCL (xGMI).
//
// The reference has no multi-GPU code; scenes shard with no exchange step, so the only
// traffic is this start-up broadcast.  RCCL is resolved at RUN TIME (dlopen of the librccl the process already has --
// PyTorch-ROCm ships one -- or the ROCm installation's), so libwjhip.so carries no link-time dependency on it and a
// single-GPU process never loads it.  whisperjav_amd/sharding.py uses torch.distributed for the same broadcast by
// default (backend "nccl" IS RCCL); these entry points are what a non-PyTorch host would bind.
#include <dlfcn.h>

#include "common.hpp"

using namespace wj;

namespace {
typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
typedef int (*fn_get_id)(rccl_unique_id*);
typedef int (*fn_init_rank)(rccl_comm_t*, int, rccl_unique_id, int);
typedef int (*fn_bcast)(const void*, void*, size_t, int /*ncclDataType_t*/, int, rccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(rccl_comm_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(rccl_comm_t, int*);

struct Rccl {
  void* lib = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_bcast bcast = nullptr;
  fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
  fn_count count = nullptr;       // ncclCommCount (optional: diagnostics)
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return WJ_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // an RCCL the process already carries under ANY soname (PyTorch bundles its own copy) is reused before anything is
  // opened: two RCCL runtimes in one process would each bootstrap their own communicators
  if (dlsym(RTLD_DEFAULT, "ncclGetUniqueId") && dlsym(RTLD_DEFAULT, "ncclCommInitRank") && dlsym(RTLD_DEFAULT, "ncclBroadcast") &&
      dlsym(RTLD_DEFAULT, "ncclCommDestroy"))
    h = dlopen(nullptr, RTLD_NOW);                  // handle of the global scope: dlsym below resolves the loaded copy
  for (const char* n : names) {
    if (h) break;
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy the process already has, if any
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) { set_error("RCCL not found (librccl.so.1): %s", dlerror()); return WJ_E_UNSUPPORTED; }
  Rccl r;
  r.lib = h;
  r.get_id = reinterpret_cast<fn_get_id>(dlsym(h, "ncclGetUniqueId"));
  r.init_rank = reinterpret_cast<fn_init_rank>(dlsym(h, "ncclCommInitRank"));
  r.bcast = reinterpret_cast<fn_bcast>(dlsym(h, "ncclBroadcast"));
  r.destroy = reinterpret_cast<fn_destroy>(dlsym(h, "ncclCommDestroy"));
  r.errstr = reinterpret_cast<fn_errstr>(dlsym(h, "ncclGetErrorString"));
  r.count = reinterpret_cast<fn_count>(dlsym(h, "ncclCommCount"));
  if (!r.get_id || !r.init_rank || !r.bcast || !r.destroy) { set_error("librccl lacks the expected nccl* symbols"); return WJ_E_UNSUPPORTED; }
  g_rccl = r;
  return WJ_OK;
}

int check_rccl(int rc, const char* what) {
  if (rc == 0) return WJ_OK;
  set_error("%s failed: %s", what, g_rccl.errstr ? g_rccl.errstr(rc) : "RCCL error");
  return WJ_E_HIP;
}
}  // namespace

struct wj_comm {
  wj_ctx* ctx = nullptr;
  rccl_comm_t comm = nullptr;
  int nranks = 0, rank = 0;
};

extern "C" {

int wj_comm_unique_id(char out[128]) {
  WJ_REQUIRE(out != nullptr, "wj_comm_unique_id: NULL argument");
  int rc = load_rccl();
  if (rc) return rc;
  rccl_unique_id id;
  rc = check_rccl(g_rccl.get_id(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(out, id.internal, 128);
  return WJ_OK;
}

int wj_comm_init(wj_ctx* ctx, int nranks, int rank, const char id[128], wj_comm** out) {
  WJ_REQUIRE(ctx && id && out && nranks >= 1 && rank >= 0 && rank < nranks, "wj_comm_init: bad arguments");
  int rc = load_rccl();
  if (rc) return rc;
  WJ_HIP(hipSetDevice(ctx->device));
  rccl_unique_id uid;
  memcpy(uid.internal, id, 128);
  wj_comm* c = new wj_comm();
  c->ctx = ctx; c->nranks = nranks; c->rank = rank;
  rc = check_rccl(g_rccl.init_rank(&c->comm, nranks, uid, rank), "ncclCommInitRank");
  if (rc) { delete c; return rc; }
  *out = c;
  return WJ_OK;
}

int wj_bcast_weights(wj_comm* comm, void* blob_dev, int64_t bytes, int root, void* stream) {
  WJ_REQUIRE(comm && blob_dev && bytes > 0 && root >= 0 && root < comm->nranks, "wj_bcast_weights: bad arguments");
  WJ_HIP(hipSetDevice(comm->ctx->device));
  hipStream_t s = comm->ctx->pick(stream);
  int rc = check_rccl(g_rccl.bcast(blob_dev, blob_dev, (size_t)bytes, /*ncclUint8*/ 1, root, comm->comm, s), "ncclBroadcast");
  if (rc) return rc;
  WJ_HIP(hipStreamSynchronize(s));
  return WJ_OK;
}

int wj_comm_count(wj_comm* comm, int* n_ranks_out) {
  WJ_REQUIRE(comm && n_ranks_out, "wj_comm_count: NULL argument");
  // what RCCL itself says about the communicator this rank built (ncclCommCount through the same dlsym table), not the number
  // the caller passed to wj_comm_init: a multi-GPU test asserts on it
  if (!g_rccl.count) { set_error("wj_comm_count: this librccl exports no ncclCommCount"); return WJ_E_UNSUPPORTED; }
  int n = 0;
  int rc = check_rccl(g_rccl.count(comm->comm, &n), "ncclCommCount");
  if (rc) return rc;
  *n_ranks_out = n;
  return WJ_OK;
}

int wj_comm_destroy(wj_comm* comm) {
  if (!comm) return WJ_OK;
  if (comm->comm && g_rccl.destroy) (void)g_rccl.destroy(comm->comm);
  delete comm;
  return WJ_OK;
}

}  // extern "C"
