// Communicator for the ONE exchange step of the path: the shared-landmark bundle-adjustment reduction of a single
// sequence whose landmarks are sharded over the GPUs of a node (BASELINE config 5).  One process per
// GPU; the collectives are RCCL's (all-reduce of the packed [Gram tiles | camera sums | max-gradient slots] and of the
// 4 step statistics per LM iteration, all-gather of the adjusted points once per adjust), enqueued on the ctx stream
// so that they order with the kernels of the iteration and need no host synchronisation.
//
// librccl is opened lazily with dlopen: a process that never creates a communicator never loads it.
#include "vo_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct rccl_api {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

rccl_api* rccl() {
  // function-local static initialised by a lambda: C++11 guarantees one thread builds the table and every other caller
  // (bench.py drives contexts from several host threads) sees it complete
  static rccl_api* const table = []() {
    rccl_api* api = new rccl_api();
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api->handle) break;
    }
    if (!api->handle) { const char* e = dlerror(); api->err = std::string("dlopen librccl: ") + (e ? e : "?"); return api; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(api->handle, n); if (!p) { ok = false; api->err = std::string("librccl lacks ") + n; } return p; };
    api->GetUniqueId = reinterpret_cast<decltype(api->GetUniqueId)>(sym("ncclGetUniqueId"));
    api->CommInitRank = reinterpret_cast<decltype(api->CommInitRank)>(sym("ncclCommInitRank"));
    api->CommDestroy = reinterpret_cast<decltype(api->CommDestroy)>(sym("ncclCommDestroy"));
    api->AllReduce = reinterpret_cast<decltype(api->AllReduce)>(sym("ncclAllReduce"));
    api->AllGather = reinterpret_cast<decltype(api->AllGather)>(sym("ncclAllGather"));
    api->GetErrorString = reinterpret_cast<decltype(api->GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(api->handle); api->handle = nullptr; }
    return api;
  }();
  return table;
}

int32_t rccl_fail(vo_ctx* c, const char* what, ncclResult_t r) {
  rccl_api* a = rccl();
  return vo_fail(c, VO_E_HIP, std::string(what) + ": " + (a->GetErrorString ? a->GetErrorString(r) : "rccl error"));
}
}  // namespace

static_assert(VO_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int32_t vo_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return VO_E_INVALID;
  rccl_api* a = rccl();
  if (!a->handle) return VO_E_STATE;
  ncclUniqueId id;
  if (a->GetUniqueId(&id) != ncclSuccess) return VO_E_HIP;
  memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return VO_OK;
}

extern "C" int32_t vo_comm_init(vo_ctx* c, int32_t n_ranks, int32_t rank, const uint8_t* id) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, id && n_ranks >= 1 && n_ranks <= VO_COMM_MAX_RANKS && rank >= 0 && rank < n_ranks, VO_E_INVALID, "bad rank / size");
  VO_CHECK(c, !c->comm, VO_E_STATE, "communicator exists");
  rccl_api* a = rccl();
  if (!a->handle) return vo_fail(c, VO_E_STATE, a->err);
  VO_HIP(c, hipSetDevice(c->device));
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = a->CommInitRank(&comm, n_ranks, uid, rank);
  if (r != ncclSuccess) return rccl_fail(c, "ncclCommInitRank", r);
  c->comm = comm; c->comm_rank = rank; c->comm_ranks = n_ranks;
  return VO_OK;
}

extern "C" int32_t vo_comm_destroy(vo_ctx* c) {
  if (!c) return VO_E_INVALID;
  if (c->comm) {
    (void)hipStreamSynchronize(c->stream);
    (void)rccl()->CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
  }
  c->comm_rank = 0; c->comm_ranks = 1;
  return VO_OK;
}

// in-place sum over the ranks, on the ctx stream; a no-op without a communicator
int32_t vo_comm_allreduce_f64(vo_ctx* c, double* buf, size_t count) {
  if (!c->comm) return VO_OK;
  const ncclResult_t r = rccl()->AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream);
  if (r != ncclSuccess) return rccl_fail(c, "ncclAllReduce", r);
  return VO_OK;
}

// recv [n_ranks][count] <- send [count] of every rank; a device copy without a communicator
int32_t vo_comm_allgather_f64(vo_ctx* c, const double* send, double* recv, size_t count) {
  if (!c->comm) {
    VO_HIP(c, hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return VO_OK;
  }
  const ncclResult_t r = rccl()->AllGather(send, recv, count, ncclDouble, static_cast<ncclComm_t>(c->comm), c->stream);
  if (r != ncclSuccess) return rccl_fail(c, "ncclAllGather", r);
  return VO_OK;
}
