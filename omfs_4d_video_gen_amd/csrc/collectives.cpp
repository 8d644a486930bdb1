// RCCL behind the C ABI: the gradient exchange of a data-parallel training step for hosts that are not PyTorch
// (inner contract: `rccl_allreduce_grads`).  RCCL is bound at run time (dlopen on first use: a
// process that never exchanges gradients neither loads nor needs librccl; a process that already holds one -- a
// PyTorch host -- gets that one), so libomfs_splat.so has no link-time dependency on it.  One communicator per rank
// and process; every call enqueues on the caller's stream and returns.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include "common.hpp"

namespace {
// the part of rccl.h this file uses (ncclResult_t 0 = success; ncclFloat32 = 7, ncclSum = 0: rccl.h, unchanged since NCCL 2)
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
struct Rccl {
  int (*GetUniqueId)(UniqueId*);
  int (*CommInitRank)(Comm*, int, UniqueId, int);
  int (*CommDestroy)(Comm);
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, hipStream_t);
  const char* (*GetErrorString)(int);
};
constexpr int kFloat32 = 7, kSum = 0;

const Rccl* rccl(const char** why) {
  static Rccl api;
  static const char* err = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"})          // a library the process already holds wins
      if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { err = "librccl.so not found (dlopen)"; return; }
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) err = n; return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.ReduceScatter = (decltype(api.ReduceScatter))sym("ncclReduceScatter");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  });
  if (why) *why = err;
  return err ? nullptr : &api;
}

int fail(const Rccl* r, const char* what, int rc) {
  return omfs::set_error(OMFS_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
}
}  // namespace

#define OMFS_RCCL(r)                                                                        \
  const char* why__ = nullptr;                                                              \
  const Rccl* r = rccl(&why__);                                                             \
  if (!r) return omfs::set_error(OMFS_ERR_HIP, "%s: RCCL unavailable: %s", __func__, why__)

extern "C" int omfs_comm_unique_id(void* id_host128) {
  OMFS_REQUIRE(id_host128, "null pointer");
  OMFS_RCCL(r);
  UniqueId id;
  if (int rc = r->GetUniqueId(&id)) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id_host128, &id, sizeof(id));
  return OMFS_OK;
}

extern "C" int omfs_comm_create(const void* id_host128, int rank, int world_size, void** comm_out) {
  OMFS_REQUIRE(id_host128 && comm_out && world_size >= 1 && rank >= 0 && rank < world_size, "arguments");
  OMFS_RCCL(r);
  UniqueId id;
  memcpy(&id, id_host128, sizeof(id));
  Comm c = nullptr;
  if (int rc = r->CommInitRank(&c, world_size, id, rank)) return fail(r, "ncclCommInitRank", rc);
  *comm_out = c;
  return OMFS_OK;
}

extern "C" int omfs_comm_destroy(void* comm) {
  if (!comm) return OMFS_OK;
  OMFS_RCCL(r);
  if (int rc = r->CommDestroy((Comm)comm)) return fail(r, "ncclCommDestroy", rc);
  return OMFS_OK;
}

extern "C" int omfs_rccl_allreduce_grads(void* comm, float* grads, size_t count, void* stream) {
  OMFS_REQUIRE(comm && grads && count > 0, "arguments");
  OMFS_RCCL(r);
  if (int rc = r->AllReduce(grads, grads, count, kFloat32, kSum, (Comm)comm, (hipStream_t)stream)) return fail(r, "ncclAllReduce", rc);
  return OMFS_OK;
}

extern "C" int omfs_rccl_allgather(void* comm, const float* mine, float* all, size_t count_per_rank, void* stream) {
  OMFS_REQUIRE(comm && mine && all && count_per_rank > 0, "arguments");
  OMFS_RCCL(r);
  if (int rc = r->AllGather(mine, all, count_per_rank, kFloat32, (Comm)comm, (hipStream_t)stream)) return fail(r, "ncclAllGather", rc);
  return OMFS_OK;
}

extern "C" int omfs_rccl_reduce_scatter(void* comm, const float* full, float* shard, size_t count_per_rank, void* stream) {
  OMFS_REQUIRE(comm && full && shard && count_per_rank > 0, "arguments");
  OMFS_RCCL(r);
  if (int rc = r->ReduceScatter(full, shard, count_per_rank, kFloat32, kSum, (Comm)comm, (hipStream_t)stream))
    return fail(r, "ncclReduceScatter", rc);
  return OMFS_OK;
}
