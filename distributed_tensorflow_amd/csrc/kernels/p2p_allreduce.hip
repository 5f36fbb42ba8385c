// One-shot all-reduce + SGD over peer-mapped buffers (one node, up to 8 GPUs on the xGMI mesh).
//
// The reference's sync step moves 318 KB of gradients per worker (X3 of its message inventory,
// /root/reference/distributed.py:97-102): far too little for a ring, whose cost at that size is the
// latency of 2(N-1) dependent hops.  On a full mesh every GPU can instead read each peer's gradient
// buffer directly over that peer's own link, all links at once, in ONE hop:
//
//   setup    every rank allocates one uncached region [3 slots x n floats | 3 x 8 flags], exports it with an IPC
//            handle and maps the regions of all peers.
//   step     the backward launch writes the rank's gradients straight into its slot (no staging copy).
//            The update kernel then (a) stores the step number into flag[slot][me] of EVERY peer
//            (system-scope release: "my slot is complete"), (b) polls its LOCAL flags until every peer's
//            step number has arrived (system-scope acquire; bounded by a wall-clock limit, never an
//            endless spin), (c) sums the N slots in rank order - identical on every rank, so the
//            replicas stay bit-identical - and applies theta -= lr/N * sum.
//   reuse    a slot is rewritten three steps later; a peer can only have published step s+1 after its
//            update kernel of step s finished reading, and this rank's update kernel of step s+2 waits
//            for those publications, so no reader is ever overtaken.
//
// The region also carries per-workgroup arrival flags [3][8][DTF_MLP2_MAX_WG] for the variant in which the
// backward launch of the two-launch step does the exchange itself (mlp2_fused.hip, mode 2).
// Used automatically for multi-rank RCCL groups on one node after a self-check (DTF_P2P_ALLREDUCE=0: RCCL).
#include <hip/hip_runtime.h>

#include <cstring>

#include "dtf_kernels.h"

namespace {

constexpr int kMaxPeers = 8;
constexpr int kSlots = 3;

struct P2PComm {
  int rank = 0, world = 0;
  int64_t n = 0;           // floats per slot (multiple of 4)
  size_t bytes = 0;
  char* local = nullptr;
  char* peer[kMaxPeers] = {};
  bool opened[kMaxPeers] = {};
  int* err = nullptr;      // device flag: 1 = a wait ran into its time limit
};

inline float* slot_of(char* base, int64_t n, int slot) { return reinterpret_cast<float*>(base) + (size_t)slot * n; }
inline int* flags_of(char* base, int64_t n) { return reinterpret_cast<int*>(base + (size_t)kSlots * n * sizeof(float)); }
inline int* wgflags_of(char* base, int64_t n) { return flags_of(base, n) + kSlots * kMaxPeers; }
constexpr size_t kFlagInts = (size_t)kSlots * kMaxPeers * (1 + DTF_MLP2_MAX_WG);
// Behind the slots and their flags, for the sharded exchange (mlp2_fused.hip, mode 4) and the general
// aggregation kernel below: inbox[8][n] | outbox[n] | inflags[8][MAX_WG] | outflags[MAX_WG] | sync words.
constexpr size_t kSyncInts = 256;           // arrival stamps, decisions, "done" marks of the general aggregation kernel
inline size_t shard_base(int64_t n) { return ((size_t)kSlots * n * sizeof(float) + kFlagInts * sizeof(int) + 255) & ~(size_t)255; }
inline size_t inbox_off(int64_t n) { return shard_base(n); }
inline size_t outbox_off(int64_t n) { return inbox_off(n) + (size_t)kMaxPeers * n * sizeof(float); }
inline size_t inflag_off(int64_t n) { return outbox_off(n) + (size_t)n * sizeof(float); }
inline size_t outflag_off(int64_t n) { return inflag_off(n) + (size_t)kMaxPeers * DTF_MLP2_MAX_WG * sizeof(int); }
inline size_t sync_off(int64_t n) { return (outflag_off(n) + (size_t)DTF_MLP2_MAX_WG * sizeof(int) + 255) & ~(size_t)255; }
inline size_t region_bytes(int64_t n) { return sync_off(n) + kSyncInts * sizeof(int); }

struct P2PKernelArgs {
  const float* data[kMaxPeers];   // slot of every rank (own rank: local pointer)
  int* flags_peer[kMaxPeers];     // flag[slot][.] array of every rank
  int* flags_local;
  float* params;
  float* sum_out;                 // optional: the reduced gradient
  const int* seq_ptr;             // device counter holding this step's number (the same on every rank)
  int* err;
  int64_t n4;                     // float4 elements
  int rank, world, slot;
  float scale;                    // lr / R
  unsigned timeout_ticks;         // of the 100 MHz constant clock
};

__global__ __launch_bounds__(256) void p2p_allreduce_sgd_kernel(P2PKernelArgs a) {
  __shared__ int timed_out;
  const int tid = threadIdx.x;
  const int seq = *a.seq_ptr;
  if (tid == 0) timed_out = 0;
  if (blockIdx.x == 0 && tid < a.world && tid != a.rank)
    __hip_atomic_store(a.flags_peer[tid] + a.slot * kMaxPeers + a.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (tid < a.world && tid != a.rank) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int* f = a.flags_local + a.slot * kMaxPeers + tid;
    // step numbers only grow: ">= seq" also accepts a peer that is already one step ahead
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
        timed_out = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (timed_out) {
    if (tid == 0) *a.err = 1;
    return;            // parameters stay untouched; the host sees the flag
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < a.n4; i += stride) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < kMaxPeers; ++p)
      if (p < a.world) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.data[p]) + i);
    f32x4* q = reinterpret_cast<f32x4*>(a.params) + i;
    *q = *q - a.scale * s;
    if (a.sum_out) reinterpret_cast<f32x4*>(a.sum_out)[i] = s;
  }
}

}  // namespace

// Creates the communicator state of this rank and writes the 64-byte IPC handle of its region to handle_out.
DTF_EXPORT void* dtf_p2p_create(int rank, int world, int64_t n_floats, char* handle_out) {
  if (rank < 0 || world < 1 || world > kMaxPeers || rank >= world || n_floats <= 0 || (n_floats & 3) || !handle_out) return nullptr;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  P2PComm* c = new P2PComm;
  c->rank = rank; c->world = world; c->n = n_floats;
  c->bytes = region_bytes(n_floats);
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, c->bytes, hipDeviceMallocUncached) != hipSuccess) { delete c; return nullptr; }
  c->local = static_cast<char*>(p);
  if (hipMemset(p, 0, c->bytes) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&c->err), sizeof(int)) != hipSuccess ||
      hipMemset(c->err, 0, sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    delete c;
    return nullptr;
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipFree(p);
    (void)hipFree(c->err);
    delete c;
    return nullptr;
  }
  memcpy(handle_out, &h, sizeof(h));
  c->peer[rank] = c->local;
  return c;
}

// all_handles: world x 64 bytes in rank order.  Maps every peer's region.
DTF_EXPORT int dtf_p2p_connect(void* comm, const char* all_handles) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c || !all_handles) return DTF_ERR_ARG;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, all_handles + (size_t)p * sizeof(h), sizeof(h));
    void* ptr = nullptr;
    if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return DTF_ERR_LAUNCH;
    c->peer[p] = static_cast<char*>(ptr);
    c->opened[p] = true;
  }
  return DTF_OK;
}

DTF_EXPORT float* dtf_p2p_slot_ptr(void* comm, int slot) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c || slot < 0 || slot >= kSlots) return nullptr;
  return slot_of(c->local, c->n, slot);
}

// Enqueue: params -= scale * sum over ranks of slot `slot`; every rank must enqueue the same (slot, *seq_ptr).
DTF_EXPORT int dtf_p2p_allreduce_sgd(void* comm, int slot, const int* seq_ptr, float* params, float* sum_out, float scale,
                                     double timeout_s, hipStream_t s) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c || !seq_ptr || !params || slot < 0 || slot >= kSlots) return DTF_ERR_ARG;
  if (!dtf_aligned16(params) || (sum_out && !dtf_aligned16(sum_out))) return DTF_ERR_ALIGN;
  P2PKernelArgs a = {};
  for (int p = 0; p < c->world; ++p) {
    if (!c->peer[p]) return DTF_ERR_ARG;      // dtf_p2p_connect has not run
    a.data[p] = slot_of(c->peer[p], c->n, slot);
    a.flags_peer[p] = flags_of(c->peer[p], c->n);
  }
  a.flags_local = flags_of(c->local, c->n);
  a.params = params; a.sum_out = sum_out; a.seq_ptr = seq_ptr; a.err = c->err;
  a.n4 = c->n >> 2; a.rank = c->rank; a.world = c->world; a.slot = slot; a.scale = scale;
  const double ticks = timeout_s * 1e8;
  a.timeout_ticks = ticks > 4.0e9 ? 4000000000u : (unsigned)ticks;
  int grid = (int)((a.n4 + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(p2p_allreduce_sgd_kernel, dim3(grid), dim3(256), 0, s, a);
  return dtf_check_launch();
}

DTF_EXPORT int dtf_p2p_update(const DtfP2PUpdateArgs* u, hipStream_t s) {
  if (!u) return DTF_ERR_ARG;
  return dtf_p2p_allreduce_sgd(u->comm, u->slot, u->seq_ptr, u->params, u->sum_out, u->scale, u->timeout_s, s);
}

DTF_EXPORT int dtf_p2p_mlp2_peer(void* comm, int slot, double timeout_s, DtfMlp2Peer* out) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c || !out || slot < 0 || slot >= kSlots) return DTF_ERR_ARG;
  memset(out, 0, sizeof(*out));
  for (int p = 0; p < c->world; ++p) {
    if (!c->peer[p]) return DTF_ERR_ARG;      // dtf_p2p_connect has not run
    out->slot[p] = slot_of(c->peer[p], c->n, slot);
    out->wgflags[p] = wgflags_of(c->peer[p], c->n);
  }
  out->err = c->err;
  out->rank = c->rank;
  out->world = c->world;
  const double ticks = timeout_s * 1e8;
  out->timeout_ticks = ticks > 4.0e9 ? 4000000000u : (unsigned)ticks;
  return DTF_OK;
}

DTF_EXPORT int dtf_p2p_mlp2_shard(void* comm, double timeout_s, DtfMlp2Shard* out) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c || !out) return DTF_ERR_ARG;
  memset(out, 0, sizeof(*out));
  for (int p = 0; p < c->world; ++p) {
    if (!c->peer[p]) return DTF_ERR_ARG;      // dtf_p2p_connect has not run
    out->region[p] = c->peer[p];
  }
  out->err = c->err;
  out->inbox_off = (long long)inbox_off(c->n);
  out->outbox_off = (long long)outbox_off(c->n);
  out->inflag_off = (long long)inflag_off(c->n);
  out->outflag_off = (long long)outflag_off(c->n);
  out->n = c->n;
  out->rank = c->rank;
  out->world = c->world;
  const double ticks = timeout_s * 1e8;
  out->timeout_ticks = ticks > 4.0e9 ? 4000000000u : (unsigned)ticks;
  return DTF_OK;
}

// 1 if any wait of this communicator ran into its time limit (synchronises with the device).
DTF_EXPORT int dtf_p2p_error(void* comm) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c) return -1;
  int e = 0;
  if (hipMemcpy(&e, c->err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return e;
}

DTF_EXPORT void dtf_p2p_destroy(void* comm) {
  P2PComm* c = static_cast<P2PComm*>(comm);
  if (!c) return;
  for (int p = 0; p < c->world; ++p)
    if (c->opened[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
  (void)hipFree(c->local);
  (void)hipFree(c->err);
  delete c;
}

// ------------------------------------------------------------------------------------------
// Asynchronous (Hogwild) push/pull over peer-mapped parameter slices.
//
// The reference's async mode (/root/reference/distributed.py:89, :102 without --sync_replicas) lets every
// worker apply its gradient to the ps-resident variables the moment it is computed, without locks, and read
// the variables back for its next step.  Here the ps-resident variables are slices of the flat parameter
// buffer living in the exported region of their owner rank; a worker applies its gradient with ONE kernel
// that walks every slice through its mapped pointer: read master, subtract lr*g, write master (the lock-free
// read-modify-write of use_locking=False, now over xGMI), and keep the fresh value as the local replica (the
// pull).  global_step is a counter in the region of shard 0, bumped with a system-scope atomic.  No host
// thread, no receive loop on the owner, nothing to wait for.
// ------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void async_push_pull_kernel(DtfAsyncPushPullArgs a) {
  const int64_t slice4 = a.slice >> 2;
  const int64_t total4 = slice4 * a.shards;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int s = (int)(i / slice4);
    f32x4* m = reinterpret_cast<f32x4*>(a.master[s]) + (i - (int64_t)s * slice4);
    f32x4 v = __builtin_nontemporal_load(m);
    if (a.mode == 0) {
      v -= a.lr * reinterpret_cast<const f32x4*>(a.grads)[i];
      __builtin_nontemporal_store(v, m);
    }
    reinterpret_cast<f32x4*>(a.params)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int gs;
    if (a.mode == 0) {
      gs = __hip_atomic_fetch_add(a.counters[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
      for (int s = 0; s < a.shards; ++s) __hip_atomic_fetch_add(a.counters[s] + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      gs = __hip_atomic_load(a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    a.state[0] = gs;      // global_step as the ps holds it after this worker's update
  }
}
}  // namespace

DTF_EXPORT int dtf_async_push_pull(const DtfAsyncPushPullArgs* a, hipStream_t s) {
  if (!a || !a->params || !a->state || a->shards < 1 || a->shards > 8 || a->slice <= 0 || (a->slice & 3)) return DTF_ERR_ARG;
  if (a->mode == 0 && !a->grads) return DTF_ERR_ARG;
  for (int i = 0; i < a->shards; ++i)
    if (!a->master[i] || !a->counters[i]) return DTF_ERR_ARG;
  if (!dtf_aligned16(a->params) || (a->grads && !dtf_aligned16(a->grads))) return DTF_ERR_ALIGN;
  const int64_t total4 = (a->slice >> 2) * a->shards;
  int grid = (int)((total4 + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(async_push_pull_kernel, dim3(grid), dim3(256), 0, s, *a);
  return dtf_check_launch();
}

// Uncached device memory that other processes of the node can map: allocation + 64-byte IPC handle.
DTF_EXPORT void* dtf_ipc_alloc(int64_t bytes, char* handle_out64) {
  if (bytes <= 0 || !handle_out64) return nullptr;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached) != hipSuccess) return nullptr;
  hipIpcMemHandle_t h;
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipFree(p);
    return nullptr;
  }
  memcpy(handle_out64, &h, sizeof(h));
  return p;
}
DTF_EXPORT void* dtf_ipc_open(const char* handle64) {
  if (!handle64) return nullptr;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return nullptr;
  return p;
}
DTF_EXPORT int dtf_ipc_close(void* p) { return p && hipIpcCloseMemHandle(p) == hipSuccess ? DTF_OK : DTF_ERR_ARG; }
DTF_EXPORT int dtf_ipc_free(void* p) { return p && hipFree(p) == hipSuccess ? DTF_OK : DTF_ERR_ARG; }
