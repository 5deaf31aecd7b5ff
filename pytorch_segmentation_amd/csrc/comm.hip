// Gradient exchange over RCCL, behind the C ABI: allreduce_bucket(flat_grad, stream).
//
// The reference trains data-parallel through torch.distributed.launch + DistributedDataParallel inside its external Trainer
// (README.md:42-44, train.py:33-35,112-117): per step, the gradients of every bucket are summed over the ranks while
// backward is still running.  Here a bucket is a contiguous range of the ONE fp32 gradient arena, so the exchange is a
// single in-place ncclAllReduce per bucket on a side stream -- no flatten / unflatten copies, no per-call Python objects.
//
// RCCL is bound at run time (dlopen: PSEG_RCCL_PATH if set, else the copy the host process already holds -- torch ships one --
// else librccl.so from the loader path), so libpseg_amd.so has no link-time dependency on it and loads on boxes without RCCL; the entry points
// below then fail with a message, nothing else does.  The communicator is the caller's to create from a 128-byte unique
// id it distributes by any channel it has (torch.distributed's store, MPI, a file): one rank calls pseg_comm_unique_id,
// every rank pseg_comm_init.
#include "common.h"

#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "../../include/pseg_amd.h"

namespace pseg {

struct RcclId {
  char internal[128];
};

typedef int (*get_unique_id_fn)(RcclId*);
typedef int (*comm_init_rank_fn)(void**, int, RcclId, int);
typedef int (*comm_destroy_fn)(void*);
typedef int (*all_reduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*reduce_scatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*all_gather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*get_version_fn)(int*);
typedef const char* (*error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_reduce_fn all_reduce = nullptr;
  reduce_scatter_fn reduce_scatter = nullptr;
  all_gather_fn all_gather = nullptr;
  get_version_fn get_version = nullptr;
  error_string_fn error_string = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;

// (bound once, whichever thread comes first: grad_ready callbacks may arrive on the autograd worker thread)
static const Rccl* rccl() {
  std::call_once(g_rccl_once, [] {
    void* h = nullptr;
    // PSEG_RCCL_PATH names the library outright (another RCCL build -- or tests/standin_rccl.cpp, the shared-memory stand-in
    // that gives two ranks on ONE device N-rank semantics); RTLD_LOCAL: its nccl* symbols must not shadow the process's own
    const char* env = getenv("PSEG_RCCL_PATH");
    if (env && env[0]) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);        // the copy the process already uses
    }
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (h) break;
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (h) {
      g_rccl.get_unique_id = (get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
      g_rccl.comm_init_rank = (comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
      g_rccl.comm_destroy = (comm_destroy_fn)dlsym(h, "ncclCommDestroy");
      g_rccl.all_reduce = (all_reduce_fn)dlsym(h, "ncclAllReduce");
      g_rccl.reduce_scatter = (reduce_scatter_fn)dlsym(h, "ncclReduceScatter");
      g_rccl.all_gather = (all_gather_fn)dlsym(h, "ncclAllGather");
      g_rccl.get_version = (get_version_fn)dlsym(h, "ncclGetVersion");
      g_rccl.error_string = (error_string_fn)dlsym(h, "ncclGetErrorString");
      if (g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce && g_rccl.reduce_scatter &&
          g_rccl.all_gather)
        g_rccl.handle = h;
    }
  });
  return g_rccl.handle ? &g_rccl : nullptr;
}

#define PSEG_RCCL_TRY(expr)                                                                              \
  do {                                                                                                   \
    const int rc_ = (expr);                                                                              \
    if (rc_ != 0) {                                                                                      \
      set_error("rccl: %s failed: %s", #expr, r->error_string ? r->error_string(rc_) : "unknown error"); \
      return PSEG_ERR_HIP;                                                                               \
    }                                                                                                    \
  } while (0)

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_comm_available(void) { return rccl() != nullptr ? 1 : 0; }

int pseg_comm_version(int* version) {
  PSEG_REQUIRE(version != nullptr, "comm_version: null pointer");
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr && r->get_version != nullptr, "comm_version: librccl.so is not loaded");
  PSEG_RCCL_TRY(r->get_version(version));
  return PSEG_OK;
}

int pseg_comm_unique_id(void* id128) {
  PSEG_REQUIRE(id128 != nullptr, "comm_unique_id: null pointer");
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr, "comm_unique_id: librccl.so could not be loaded (set PSEG_RCCL_PATH)");
  RcclId id;
  PSEG_RCCL_TRY(r->get_unique_id(&id));
  memcpy(id128, id.internal, sizeof(id.internal));
  return PSEG_OK;
}

int pseg_comm_init(const void* id128, int nranks, int rank, int64_t* comm) {
  PSEG_REQUIRE(id128 != nullptr && comm != nullptr && nranks >= 1 && rank >= 0 && rank < nranks,
               "comm_init: bad argument (nranks %d rank %d)", nranks, rank);
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr, "comm_init: librccl.so could not be loaded (set PSEG_RCCL_PATH)");
  RcclId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  void* c = nullptr;
  PSEG_RCCL_TRY(r->comm_init_rank(&c, nranks, id, rank));    // on the calling thread's current HIP device
  *comm = (int64_t)(intptr_t)c;
  return PSEG_OK;
}

int pseg_comm_destroy(int64_t comm) {
  if (comm == 0) return PSEG_OK;
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr, "comm_destroy: librccl.so is not loaded");
  PSEG_RCCL_TRY(r->comm_destroy((void*)(intptr_t)comm));
  return PSEG_OK;
}

int pseg_allreduce_bucket(int64_t comm, float* flat_grad, int64_t count, void* stream) {
  PSEG_REQUIRE(comm != 0 && flat_grad != nullptr && count > 0, "allreduce_bucket: bad argument");
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr, "allreduce_bucket: librccl.so is not loaded");
  // in place, fp32 (ncclFloat32 = 7), sum (ncclSum = 0): the 1/world of the mean is folded into the optimiser's grad_scale
  PSEG_RCCL_TRY(r->all_reduce(flat_grad, flat_grad, (size_t)count, 7, 0, (void*)(intptr_t)comm, (hipStream_t)stream));
  return PSEG_OK;
}

// The same sum as two collectives: every rank first receives the sum of ITS 1/nranks slice of the bucket (reduce-scatter),
// then the reduced slices are handed round (all-gather).  On the fully connected xGMI node each of the two steps is a direct
// exchange over all seven links of a GPU -- bytes per link 2 x (n - 1) / n x bucket / 7 -- where a ring all-reduce is bound by
// ONE link; which of the two RCCL's own ncclAllReduce picks for a given size is its decision (NCCL_ALGO), this pair makes it
// the caller's.  In place: slice r of the bucket is [r * count_per_rank, (r + 1) * count_per_rank).
int pseg_reduce_scatter_bucket(int64_t comm, float* flat_grad, int64_t count_per_rank, int rank, void* stream) {
  PSEG_REQUIRE(comm != 0 && flat_grad != nullptr && count_per_rank > 0 && rank >= 0, "reduce_scatter_bucket: bad argument");
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr, "reduce_scatter_bucket: librccl.so is not loaded");
  PSEG_RCCL_TRY(r->reduce_scatter(flat_grad, flat_grad + (long long)rank * count_per_rank, (size_t)count_per_rank, 7, 0,
                                  (void*)(intptr_t)comm, (hipStream_t)stream));
  return PSEG_OK;
}

int pseg_all_gather_bucket(int64_t comm, float* flat_grad, int64_t count_per_rank, int rank, void* stream) {
  PSEG_REQUIRE(comm != 0 && flat_grad != nullptr && count_per_rank > 0 && rank >= 0, "all_gather_bucket: bad argument");
  const Rccl* r = rccl();
  PSEG_REQUIRE(r != nullptr, "all_gather_bucket: librccl.so is not loaded");
  PSEG_RCCL_TRY(r->all_gather(flat_grad + (long long)rank * count_per_rank, flat_grad, (size_t)count_per_rank, 7,
                              (void*)(intptr_t)comm, (hipStream_t)stream));
  return PSEG_OK;
}

}  // extern "C"
