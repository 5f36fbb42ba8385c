// Native RCCL communicator layer + the collective / optimizer entries of the step schedule.
//
// The cross-process traffic of the reference (parameter pulls and gradient pushes through gRPC
// RecvTensor, /root/reference/distributed.py:54, :102) becomes, per synchronous step, one
// all-reduce (small models) or one reduce-scatter + one all-gather (large models) issued on the
// SAME HIP stream as the compute kernels, by the same native call that launches them: there is no
// per-step Python, no stream hop and no host synchronisation between compute and communication.
//
// RCCL is loaded with dlopen from the path the Python layer passes in (the copy PyTorch ships, so a
// process never holds two RCCL implementations); no RCCL header is needed.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "dtf_kernels.h"

namespace {
typedef void* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { kNcclSum = 0, kNcclInt32 = 2, kNcclFloat32 = 7 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(ncclComm_t, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

template <typename F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  return fn != nullptr;
}
}  // namespace

// Returns 0 on success, 1 if the library cannot be opened, 2 if a symbol is missing.
DTF_EXPORT int dtf_comm_load(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.handle) return 0;
  g_rccl.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!g_rccl.handle) return 1;
  bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
            bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.CommCount, "ncclCommCount") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
            bind(g_rccl.ReduceScatter, "ncclReduceScatter") && bind(g_rccl.AllGather, "ncclAllGather") &&
            bind(g_rccl.Broadcast, "ncclBroadcast") && bind(g_rccl.GetErrorString, "ncclGetErrorString");
  if (!ok) {
    dlclose(g_rccl.handle);
    g_rccl = Rccl();
    return 2;
  }
  return 0;
}

DTF_EXPORT int dtf_comm_unique_id(char* out128) {
  if (!g_rccl.handle || !out128) return -1;
  ncclUniqueId id;
  const int rc = g_rccl.GetUniqueId(&id);
  memcpy(out128, id.internal, 128);
  return rc;
}

// Collective over all ranks: every rank calls this with the same id.  The calling thread's current
// HIP device is the rank's device.
DTF_EXPORT void* dtf_comm_init(const char* id128, int nranks, int rank, int* rc_out) {
  if (!g_rccl.handle || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) {
    if (rc_out) *rc_out = -1;
    return nullptr;
  }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  ncclComm_t comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, nranks, id, rank);
  if (rc_out) *rc_out = rc;
  return rc == 0 ? comm : nullptr;
}

DTF_EXPORT int dtf_comm_destroy(void* comm) { return (g_rccl.handle && comm) ? g_rccl.CommDestroy(comm) : -1; }
// Number of ranks of the communicator as RCCL itself reports it (-1 on error).
DTF_EXPORT int dtf_comm_count(void* comm) {
  int n = -1;
  if (!g_rccl.handle || !comm || g_rccl.CommCount(comm, &n) != 0) return -1;
  return n;
}
DTF_EXPORT const char* dtf_comm_error_string(int rc) { return g_rccl.handle ? g_rccl.GetErrorString(rc) : "rccl not loaded"; }

DTF_EXPORT int dtf_comm_all_reduce_f32(void* comm, float* buf, int64_t count, hipStream_t s) {
  if (!comm || !buf || count <= 0) return -1;
  return g_rccl.AllReduce(buf, buf, (size_t)count, kNcclFloat32, kNcclSum, comm, s);
}
// recv receives this rank's slice of the element-wise sum; send holds nranks * recvcount elements
DTF_EXPORT int dtf_comm_reduce_scatter_f32(void* comm, const float* send, float* recv, int64_t recvcount, hipStream_t s) {
  if (!comm || !send || !recv || recvcount <= 0) return -1;
  return g_rccl.ReduceScatter(send, recv, (size_t)recvcount, kNcclFloat32, kNcclSum, comm, s);
}
DTF_EXPORT int dtf_comm_all_gather_f32(void* comm, const float* send, float* recv, int64_t sendcount, hipStream_t s) {
  if (!comm || !send || !recv || sendcount <= 0) return -1;
  return g_rccl.AllGather(send, recv, (size_t)sendcount, kNcclFloat32, comm, s);
}
DTF_EXPORT int dtf_comm_broadcast_bytes(void* comm, void* buf, int64_t nbytes, int root, hipStream_t s) {
  if (!comm || !buf || nbytes <= 0) return -1;
  return g_rccl.Broadcast(buf, buf, (size_t)nbytes, /*ncclInt8*/ 0, root, comm, s);
}

// --------------------------------------------------------------------------------------
// schedule entries that follow the compute kernels of a step
// --------------------------------------------------------------------------------------
__global__ void int_add_kernel(int* p, int v) { *p += v; }
DTF_EXPORT int dtf_int_add(int* p, int v, hipStream_t s) {
  if (!p) return DTF_ERR_ARG;
  hipLaunchKernelGGL(int_add_kernel, dim3(1), dim3(1), 0, s, p, v);
  return dtf_check_launch();
}

DTF_EXPORT int dtf_run_update(const DtfUpdateArgs* a, hipStream_t s) {
  if (!a || !a->params || !a->grads || a->total <= 0 || a->shard_elems <= 0) return DTF_ERR_ARG;
  if (a->shard_offset < 0 || a->shard_offset + a->shard_elems > a->total) return DTF_ERR_SHAPE;
  float* pshard = a->params + a->shard_offset;
  int rc;
  if (a->mode == DTF_UPDATE_LOCAL) {
    // one rank: the gradients are already complete
    if (a->optimizer == 0) return dtf_sgd_apply(pshard, a->grads + a->shard_offset, a->shard_elems, a->lr, a->grad_scale, s);
    rc = dtf_adam_apply(pshard, a->grads + a->shard_offset, a->m, a->v, a->shard_elems, a->lr, a->beta1, a->beta2, a->eps,
                        a->grad_scale, a->step, 1, s);
    return rc ? rc : dtf_int_add(a->step, 1, s);
  }
  if (!a->comm) return DTF_ERR_ARG;
  if (a->mode == DTF_UPDATE_ALLREDUCE) {
    // latency path (small models, stateless SGD): one collective, every rank updates its full replica
    if (a->optimizer != 0) return DTF_ERR_ARG;
    if (dtf_comm_all_reduce_f32(a->comm, a->grads, a->total, s)) return DTF_ERR_LAUNCH;
    return dtf_sgd_apply(a->params, a->grads, a->total, a->lr, a->grad_scale, s);
  }
  if (a->mode == DTF_UPDATE_SHARDED) {
    // parameter-server path: sum lands on the owner of each slice, the owner applies the optimizer
    // (its state exists only for that slice), the updated slices are gathered back into every replica
    if (!a->rs_out) return DTF_ERR_ARG;
    if (dtf_comm_reduce_scatter_f32(a->comm, a->grads, a->rs_out, a->shard_elems, s)) return DTF_ERR_LAUNCH;
    if (a->optimizer == 0) {
      rc = dtf_sgd_apply(pshard, a->rs_out, a->shard_elems, a->lr, a->grad_scale, s);
    } else {
      rc = dtf_adam_apply(pshard, a->rs_out, a->m, a->v, a->shard_elems, a->lr, a->beta1, a->beta2, a->eps, a->grad_scale,
                          a->step, 1, s);
      if (!rc) rc = dtf_int_add(a->step, 1, s);
    }
    if (rc) return rc;
    return dtf_comm_all_gather_f32(a->comm, pshard, a->params, a->shard_elems, s) ? DTF_ERR_LAUNCH : DTF_OK;
  }
  return DTF_ERR_ARG;
}
