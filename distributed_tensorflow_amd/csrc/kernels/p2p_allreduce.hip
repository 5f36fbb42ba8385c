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
#include <mutex>
#include <unordered_map>
#include <vector>

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
  int* err = nullptr;      // device ints: [0] 1 = a wait ran into its time limit, [1] block counter of the general kernel
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

// Every wait below gives up at once when the error word is already set: after the first wait of a run has hit its
// limit, the remaining steps of the same graph (up to 2 x 64 launches) must not each wait that long again.
__device__ __forceinline__ bool p2p_already_failed(const int* err) {
  return __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

__global__ __launch_bounds__(256) void p2p_allreduce_sgd_kernel(P2PKernelArgs a) {
  __shared__ int timed_out;
  const int tid = threadIdx.x;
  const int seq = *a.seq_ptr;
  const bool dead = p2p_already_failed(a.err);
  if (tid == 0) timed_out = 0;
  if (blockIdx.x == 0 && tid < a.world && tid != a.rank)
    __hip_atomic_store(a.flags_peer[tid] + a.slot * kMaxPeers + a.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  if (tid < a.world && tid != a.rank) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int* f = a.flags_local + a.slot * kMaxPeers + tid;
    // step numbers only grow: ">= seq" also accepts a peer that is already one step ahead
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
      if (dead || __builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
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

// Exported (uncached, IPC-mapped) regions are never returned to the driver while the process lives: a released region
// goes to a free list and the next request of at most its size takes it again, zeroed.  Reason: device memory that had
// been such a region, was freed, and then came back through an ordinary allocation was seen holding zeros in single
// 128-byte lines no matter how often they were rewritten (the same lines on every machine tried: a program built
// right after a one-rank peer-mapped engine had been torn down found parameters and labels missing).  A process
// builds one or two engines in its life; only test processes build dozens.
namespace {
struct PooledRegion { void* p; size_t bytes; };
std::mutex g_pool_mu;
std::vector<PooledRegion> g_pool;
std::unordered_map<void*, size_t> g_ipc_sizes;      // regions handed out by dtf_ipc_alloc

void* region_acquire(size_t bytes) {
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); ++i)
      if (g_pool[i].bytes >= bytes && (best == g_pool.size() || g_pool[i].bytes < g_pool[best].bytes)) best = i;
    if (best != g_pool.size()) {
      p = g_pool[best].p;
      g_pool.erase(g_pool.begin() + (long)best);
    }
  }
  if (!p && hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;                                   // (the region stays allocated; nothing else will use it)
  }
  return p;
}

void region_release(void* p, size_t bytes) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pool.push_back({p, bytes});
}
}  // namespace

// Creates the communicator state of this rank and writes the 64-byte IPC handle of its region to handle_out.
DTF_EXPORT void* dtf_p2p_create(int rank, int world, int64_t n_floats, char* handle_out) {
  if (rank < 0 || world < 1 || world > kMaxPeers || rank >= world || n_floats <= 0 || (n_floats & 3) || !handle_out) return nullptr;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  P2PComm* c = new P2PComm;
  c->rank = rank; c->world = world; c->n = n_floats;
  c->bytes = region_bytes(n_floats);
  void* p = region_acquire(c->bytes);
  if (!p) { delete c; return nullptr; }
  c->local = static_cast<char*>(p);
  if (hipMalloc(reinterpret_cast<void**>(&c->err), 4 * sizeof(int)) != hipSuccess ||
      hipMemset(c->err, 0, 4 * sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    region_release(p, c->bytes);
    delete c;
    return nullptr;
  }
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    region_release(p, c->bytes);
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

// ------------------------------------------------------------------------------------------
// SyncReplicas with replicas_to_aggregate != number of workers, on the device
// (/root/reference/distributed.py:29-32, :92-100: ConditionalAccumulator + token queue).
//
// One launch per gradient contribution ("round"), after the backward launch has written the gradients into
// the round's slot.  Round number c = the worker's local step, equal on every rank, and the stamp a
// contribution carries:
//   1. every rank stores its stamp into arrive[slot][rank] of the CHIEF's region (rank 0 hosts the accumulator);
//   2. the chief takes the first `need` ranks whose stamp equals c - in the order its polls see them, lower
//      rank first within one poll; a stamp older than c is stale and is never taken - and stores
//      {c, accepted-rank mask} as ONE 8-byte word into decision[slot] of every rank's region;
//   3. every rank waits for that word, adds the slots of the accepted ranks in rank order (identical replicas),
//      accumulates over the rounds of an update and, in the update's last round, applies
//      theta -= lr/R * sum and bumps global_step; a rank that finds itself outside the mask counts its
//      gradient as dropped (device counter, also written to the step's log record);
//   4. every rank stores c into done[rank] of every region, then waits until every rank has finished round
//      c - 2: slots are reused every third round, so a rank may run at most two rounds ahead of the slowest
//      (the reference's token queue is unbounded; ours holds two tokens per worker).
// R < N: need = R, one round per update.  R > N: ceil(R/N) rounds per update, need = N except in the last.
// No host involvement: graph-replayable.  Every wait is bounded by a wall-clock limit.
// ------------------------------------------------------------------------------------------
namespace {
constexpr int kArriveOff = 0;        // int arrive[kSlots][kMaxPeers]
constexpr int kDecisionOff = 32;     // 8-byte decision[kSlots] (ints 32..37)
constexpr int kDoneOff = 48;         // int done[kMaxPeers]
inline int* sync_of(char* base, int64_t n) { return reinterpret_cast<int*>(base + sync_off(n)); }

struct P2PGeneralArgs {
  const float* data[kMaxPeers];   // this round's slot of every rank
  int* sync_peer[kMaxPeers];      // sync words of every rank's region
  int* sync_local;
  float* params;
  float* acc;                     // local accumulator over the rounds of an update [n]
  int* state;                     // local counters: [0] global_step, [3] gradients of this rank that were dropped
  int* wlog;                      // optional wide log (DTF_WLOG_INTS ints per record): [2] accepted mask, [3] dropped so far
  const int* seq_ptr;             // device counter holding the round number c
  int* block_ctr;                 // local, monotonic: blocks that finished their share (last-block detection)
  int* err;
  int64_t n4;
  int rank, world, slot;
  int need, first_round, last_round;
  int stamp_offset;               // test hook: the stamp published is c + stamp_offset (a stale contribution)
  int log_cap;
  unsigned delay_ticks;           // test hook: this rank's contribution arrives late by this much
  unsigned timeout_ticks;
  float scale;                    // lr / R
};

__device__ __forceinline__ bool p2p_wait_expired(unsigned long long t0, unsigned limit) {
  return __builtin_amdgcn_s_memrealtime() - t0 > limit;
}

__global__ __launch_bounds__(256) void sync_general_kernel(P2PGeneralArgs a) {
  __shared__ int sh_mask;
  __shared__ int sh_timeout;
  const int tid = threadIdx.x;
  const int seq = *a.seq_ptr;
  const bool dead = p2p_already_failed(a.err);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    sh_timeout = 0;
    sh_mask = 0;
    if (blockIdx.x == 0) {
      if (a.delay_ticks) {
        while (__builtin_amdgcn_s_memrealtime() - t0 < a.delay_ticks) __builtin_amdgcn_s_sleep(32);
      }
      // 1. my contribution (gradients complete: written by the launch before this one) and its stamp
      __hip_atomic_store(a.sync_peer[0] + kArriveOff + a.slot * kMaxPeers + a.rank, seq + a.stamp_offset, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.rank == 0) {
        // 2. the accumulator: first `need` fresh arrivals
        int mask = 0, count = 0;
        bool late = false;
        while (count < a.need) {
          for (int k = 0; k < a.world && count < a.need; ++k) {
            if (mask >> k & 1) continue;
            const int v = __hip_atomic_load(a.sync_local + kArriveOff + a.slot * kMaxPeers + k, __ATOMIC_ACQUIRE,
                                            __HIP_MEMORY_SCOPE_SYSTEM);
            if (v == seq) { mask |= 1 << k; ++count; }
          }
          if (count >= a.need) break;
          if (dead || p2p_wait_expired(t0, a.timeout_ticks)) { late = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        if (!late) {
          const unsigned long long word = ((unsigned long long)(unsigned)seq << 32) | (unsigned)mask;
          for (int k = 0; k < a.world; ++k)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.sync_peer[k] + kDecisionOff) + a.slot, word,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    // 3. the decision of this round
    const unsigned long long* d = reinterpret_cast<const unsigned long long*>(a.sync_local + kDecisionOff) + a.slot;
    for (;;) {
      const unsigned long long w = __hip_atomic_load(d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((int)(w >> 32) == seq) { sh_mask = (int)(w & 0xffffffffu); break; }
      if (dead || p2p_wait_expired(t0, a.timeout_ticks)) { sh_timeout = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (sh_timeout) {
    if (tid == 0) *a.err = 1;
    return;            // parameters stay untouched; the host sees the flag
  }
  const int mask = sh_mask;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + tid; i < a.n4; i += stride) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < kMaxPeers; ++p)
      if (p < a.world && (mask >> p & 1)) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.data[p]) + i);
    f32x4* ac = reinterpret_cast<f32x4*>(a.acc) + i;
    if (!a.first_round) s += *ac;
    if (a.last_round) {
      f32x4* q = reinterpret_cast<f32x4*>(a.params) + i;
      *q = *q - a.scale * s;
    } else {
      *ac = s;
    }
  }
  // 4. bookkeeping by the block that finishes last
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid != 0) return;
  const int old = __hip_atomic_fetch_add(a.block_ctr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if ((old + 1) % (int)gridDim.x != 0) return;
  if (!(mask >> a.rank & 1)) a.state[3] += 1;
  if (a.last_round) a.state[0] += 1;               // one global_step increment per aggregated update
  if (a.wlog) {
    int* rec = a.wlog + (size_t)DTF_WLOG_INTS * ((seq - 1) % a.log_cap);
    rec[2] = mask;
    rec[3] = a.state[3];
  }
  for (int k = 0; k < a.world; ++k)
    __hip_atomic_store(a.sync_peer[k] + kDoneOff + a.rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int k = 0; k < a.world; ++k) {
    while (__hip_atomic_load(a.sync_local + kDoneOff + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq - 2) {
      if (dead || p2p_wait_expired(t0, a.timeout_ticks)) { *a.err = 1; return; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
}  // namespace

DTF_EXPORT int dtf_p2p_sync_general(const DtfP2PGeneralArgs* u, hipStream_t s) {
  if (!u) return DTF_ERR_ARG;
  P2PComm* c = static_cast<P2PComm*>(u->comm);
  if (!c || !u->seq_ptr || !u->params || !u->acc || !u->state || u->slot < 0 || u->slot >= kSlots) return DTF_ERR_ARG;
  if (u->need < 1 || u->need > c->world || u->scale != u->scale) return DTF_ERR_ARG;
  if (!dtf_aligned16(u->params) || !dtf_aligned16(u->acc)) return DTF_ERR_ALIGN;
  if (u->wlog && u->log_cap <= 0) return DTF_ERR_ARG;
  P2PGeneralArgs a = {};
  for (int p = 0; p < c->world; ++p) {
    if (!c->peer[p]) return DTF_ERR_ARG;      // dtf_p2p_connect has not run
    a.data[p] = slot_of(c->peer[p], c->n, u->slot);
    a.sync_peer[p] = sync_of(c->peer[p], c->n);
  }
  a.sync_local = sync_of(c->local, c->n);
  a.params = u->params; a.acc = u->acc; a.state = u->state; a.wlog = u->wlog; a.seq_ptr = u->seq_ptr;
  a.block_ctr = c->err + 1; a.err = c->err;
  a.n4 = c->n >> 2; a.rank = c->rank; a.world = c->world; a.slot = u->slot;
  a.need = u->need; a.first_round = u->first_round; a.last_round = u->last_round; a.stamp_offset = u->stamp_offset;
  a.log_cap = u->log_cap; a.scale = u->scale;
  const double ticks = u->timeout_s * 1e8, dticks = u->delay_s * 1e8;
  a.timeout_ticks = ticks > 4.0e9 ? 4000000000u : (unsigned)ticks;
  a.delay_ticks = dticks > 4.0e9 ? 4000000000u : (dticks > 0 ? (unsigned)dticks : 0u);
  // the grid size is fixed per communicator: the last-block detection counts in multiples of it
  int grid = (int)((a.n4 + 255) / 256);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(sync_general_kernel, dim3(grid), dim3(256), 0, s, a);
  return dtf_check_launch();
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
  region_release(c->local, c->bytes);
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
// Staleness bound: a worker pulls on the local steps divisible by `pull_every` (1 = every step) and pushes on every
// step; between pulls it applies its own update to its replica, so the replica misses only the OTHER workers'
// updates since its last pull.  How many those were when a gradient was pushed is logged per step.
// Adam: the moments live next to the master slice in the owner's region (the slot variables of a ps-resident
// optimizer) and are updated with the same lock-free read-modify-write; the bias correction uses the number of
// updates the slice has received.
__global__ __launch_bounds__(256) void async_push_pull_kernel(DtfAsyncPushPullArgs a) {
  const int64_t slice4 = a.slice >> 2;
  const int64_t total4 = slice4 * a.shards;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int ls = a.state[1];                     // local step, already advanced by the step's tail
  // (the forced pull was decided by the previous push of this worker: one value for every workgroup of this launch)
  const bool forced = a.astate && a.bound >= 0 && a.mode == 0 && a.astate[4] != 0;
  const bool do_pull = a.mode == 1 || a.pull_every <= 1 || (ls % a.pull_every) == 0 || forced;
  int cur_s = -1;
  float lr_t = a.lr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int s = (int)(i / slice4);
    const int64_t off = i - (int64_t)s * slice4;
    f32x4* m = reinterpret_cast<f32x4*>(a.master[s]) + off;
    f32x4 v = __builtin_nontemporal_load(m);
    if (a.mode == 0) {
      const f32x4 g = reinterpret_cast<const f32x4*>(a.grads)[i];
      f32x4 nv;
      if (a.optimizer == 0) {
        nv = v - a.lr * g;
      } else {
        if (s != cur_s) {
          cur_s = s;
          const float t = (float)(__hip_atomic_load(a.counters[s] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1);
          lr_t = a.lr * sqrtf(1.f - powf(a.b2, t)) / (1.f - powf(a.b1, t));
        }
        f32x4* mp = reinterpret_cast<f32x4*>(a.m[s]) + off;
        f32x4* vp = reinterpret_cast<f32x4*>(a.v[s]) + off;
        f32x4 mm = __builtin_nontemporal_load(mp), vv = __builtin_nontemporal_load(vp);
        mm = a.b1 * mm + (1.f - a.b1) * g;
        vv = a.b2 * vv + (1.f - a.b2) * (g * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)      // hardware square root and reciprocal, as in dtf_adam_one
          nv[j] = v[j] - lr_t * mm[j] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv[j]) + a.eps);
        __builtin_nontemporal_store(mm, mp);
        __builtin_nontemporal_store(vv, vp);
      }
      __builtin_nontemporal_store(nv, m);
      if (!do_pull) {
        // push only: the replica takes this worker's own update and keeps missing the others'
        f32x4* pl = reinterpret_cast<f32x4*>(a.params) + i;
        *pl = *pl + (nv - v);
        continue;
      }
      v = nv;
    }
    reinterpret_cast<f32x4*>(a.params)[i] = v;
  }
  // The counters are advanced by the workgroup that finishes LAST: every workgroup of this launch has read the
  // slices' update counts (Adam's bias correction) by then, so a push uses one count throughout, whatever the order in
  // which its workgroups run.  (Without the bookkeeping words - the start-up check, plain SGD - workgroup 0 does it.)
  bool keeper = blockIdx.x == 0 && threadIdx.x == 0;
  if (a.astate) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      s_last = __hip_atomic_fetch_add(a.astate + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
      if (s_last) a.astate[8] = 0;
    }
    __syncthreads();
    keeper = s_last && threadIdx.x == 0;
  }
  if (keeper) {
    int gs;
    if (a.mode == 0) {
      gs = __hip_atomic_fetch_add(a.counters[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
      for (int s = 0; s < a.shards; ++s) __hip_atomic_fetch_add(a.counters[s] + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.astate) {
        // astate[0] = global_step at this worker's last pull, astate[1] = its own pushes since then
        const int unseen = gs - 1 - (a.astate[0] + a.astate[1]);
        if (a.alog && a.alog_cap > 0) {
          int* rec = a.alog + 2 * (size_t)((ls > 0 ? ls - 1 : 0) % a.alog_cap);
          rec[0] = gs;
          rec[1] = unseen < 0 ? 0 : unseen;
        }
        if (do_pull) { a.astate[0] = gs; a.astate[1] = 0; }
        else a.astate[1] += 1;
        if (a.bound >= 0) {
          // Bounded staleness (/root/reference/distributed.py:89, :102 has none: plain Hogwild).  (1) A replica that
          // has fallen more than `bound` updates of other workers behind is refreshed with the next push, whatever
          // the pull period says.  (2) Stale-synchronous clocks: every worker publishes the number of steps it has
          // pushed; a worker more than `bound` steps ahead of the slowest waits here, i.e. before its next step
          // starts.  A worker that has finished parks its clock at the maximum; a wait that runs into its limit (a
          // worker died) gives the bound up for the rest of the run instead of stalling again at every step.
          if (forced) a.astate[2] += 1;
          a.astate[4] = (!do_pull && unseen >= (a.bound > 0 ? a.bound : 1)) ? 1 : 0;
          int* clocks = a.counters[0] + 4;
          __hip_atomic_store(clocks + a.rank, ls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (a.astate[5] == 0 && a.world > 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool waited = false;
            for (;;) {
              int slowest = 0x7fffffff;
              for (int j = 0; j < a.world; ++j)
                if (j != a.rank) slowest = min(slowest, __hip_atomic_load(clocks + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
              if (ls - slowest <= a.bound) break;
              waited = true;
              if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                a.astate[5] = 1;
                break;
              }
              __builtin_amdgcn_s_sleep(16);
            }
            if (waited) {
              a.astate[3] += 1;
              a.astate[4] = 1;                   // the others pushed while this worker waited: refresh with the next push
            }
          }
        }
      }
    } else {
      gs = __hip_atomic_load(a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (a.astate) { a.astate[0] = gs; a.astate[1] = 0; }
    }
    a.state[0] = gs;      // global_step as the ps holds it after this worker's update
  }
}
}  // namespace

DTF_EXPORT int dtf_async_push_pull(const DtfAsyncPushPullArgs* a, hipStream_t s) {
  if (!a || !a->params || !a->state || a->shards < 1 || a->shards > 8 || a->slice <= 0 || (a->slice & 3)) return DTF_ERR_ARG;
  if (a->mode == 0 && !a->grads) return DTF_ERR_ARG;
  if (a->optimizer != 0 && a->optimizer != 1) return DTF_ERR_ARG;
  if (a->alog && !a->astate) return DTF_ERR_ARG;
  if (a->bound >= 0 && (!a->astate || a->world < 1 || a->world > 8 || a->rank < 0 || a->rank >= a->world)) return DTF_ERR_ARG;
  for (int i = 0; i < a->shards; ++i) {
    if (!a->master[i] || !a->counters[i]) return DTF_ERR_ARG;
    if (a->optimizer == 1 && a->mode == 0 && (!a->m[i] || !a->v[i])) return DTF_ERR_ARG;
  }
  if (!dtf_aligned16(a->params) || (a->grads && !dtf_aligned16(a->grads))) return DTF_ERR_ALIGN;
  const int64_t total4 = (a->slice >> 2) * a->shards;
  // enough workgroups to keep the memory system busy on the 100 M-parameter model, one per 256 float4 on the small one
  int64_t grid = (total4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(async_push_pull_kernel, dim3((int)grid), dim3(256), 0, s, *a);
  return dtf_check_launch();
}

// Uncached device memory that other processes of the node can map: allocation + 64-byte IPC handle.
DTF_EXPORT void* dtf_ipc_alloc(int64_t bytes, char* handle_out64) {
  if (bytes <= 0 || !handle_out64) return nullptr;
  void* p = region_acquire((size_t)bytes);
  if (!p) return nullptr;
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    region_release(p, (size_t)bytes);
    return nullptr;
  }
  memcpy(handle_out64, &h, sizeof(h));
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_ipc_sizes[p] = (size_t)bytes;
  }
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
DTF_EXPORT int dtf_ipc_free(void* p) {
  if (!p) return DTF_ERR_ARG;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_ipc_sizes.find(p);
    if (it == g_ipc_sizes.end()) return DTF_ERR_ARG;
    bytes = it->second;
    g_ipc_sizes.erase(it);
  }
  region_release(p, bytes);                         // kept for the next engine of this process, see region_acquire
  return DTF_OK;
}
