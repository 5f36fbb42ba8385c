// What does a dependent phase change cost on this chip: a kernel boundary, or a grid-wide barrier inside ONE
// persistent launch?  The reference training step is two dependent phases (forward -> backward/update -> next
// forward); each costs one kernel boundary today.  A persistent kernel would replace the boundaries by device-wide
// barriers.  This probe times both on the same skeleton: every phase, every workgroup writes its 1.3 KB share of a
// 318 KB buffer (the size of the model's parameters) and in the next phase reads a share another workgroup wrote
// (so the data really has to cross CUs / XCDs), checks it, and goes on.
//
//   launches      two kernels per step, as a hipGraph of `steps` steps (what the framework does)
//   persistent    one launch, `G` co-resident workgroups, two barriers per step:
//                   flat     one monotonic counter, agent-scope release before arriving, sc1 polling, acquire after
//                   xcd      per-XCD counters, the XCD's last arriver goes to the top counter, per-XCD release word
//   one-xcd       the same persistent kernel but only the workgroups that land on ONE XCD take part (the grid is
//                 8 x larger and the others leave at once): producer and consumer share an L2
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/phase_probe.hip -o tools/micro/phase_probe
// Run:   tools/micro/phase_probe [workgroups=104] [steps=200] [repeats=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kThreads = 256;
constexpr int kFloatsPerWG = 320;          // ~1.3 KB per workgroup per phase (x 2 buffers)
constexpr unsigned kSpinLimit = 1u << 24;  // every wait is bounded: a broken barrier ends the kernel with an error flag

struct Sync {
  unsigned flat;           // flat barrier: arrivals (monotonic)
  unsigned pad0[31];
  unsigned top;            // hierarchical: XCDs that arrived (monotonic)
  unsigned pad1[31];
  unsigned xcd_arrive[8][32];   // per-XCD arrivals, one 128-byte line each
  unsigned xcd_release[8][32];  // per-XCD generation
  unsigned census[8];      // workgroups per XCD (filled by the census kernel)
  int error;
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// one phase of work: write my share, tagged with (phase, wg)
__device__ __forceinline__ void produce(float* buf, int wg, int phase) {
  for (int i = threadIdx.x; i < kFloatsPerWG; i += kThreads)
    __hip_atomic_store(buf + (size_t)wg * kFloatsPerWG + i, (float)(phase * 1000 + wg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and read the share of workgroup (wg + stride) % G written in the previous phase
__device__ __forceinline__ int consume(const float* buf, int wg, int G, int stride, int phase) {
  const int src = (wg + stride) % G;
  int bad = 0;
  for (int i = threadIdx.x; i < kFloatsPerWG; i += kThreads) {
    const float v = __hip_atomic_load(buf + (size_t)src * kFloatsPerWG + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bad |= v != (float)(phase * 1000 + src);
  }
  return bad;
}

__global__ void phase_a(float* b0, const float* b1, int G, int step, int* err) {
  int bad = step > 0 ? consume(b1, blockIdx.x, G, 37, 2 * step - 1) : 0;
  produce(b0, blockIdx.x, 2 * step);
  if (bad) *err = 1;
}
__global__ void phase_b(const float* b0, float* b1, int G, int step, int* err) {
  int bad = consume(b0, blockIdx.x, G, 37, 2 * step);
  produce(b1, blockIdx.x, 2 * step + 1);
  if (bad) *err = 1;
}

__device__ __forceinline__ bool flat_barrier(Sync* s, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its write-through stores have been acknowledged
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&s->flat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > kSpinLimit) { ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return ok;
}

// hierarchical: my XCD's counter; the XCD's last arriver reports to the top counter, waits for every XCD, then
// bumps the generation word of every XCD; everybody else polls its own XCD's word (a line in its own L2's reach)
__device__ __forceinline__ bool xcd_barrier(Sync* s, unsigned gen, unsigned xcd, unsigned n_xcd_active) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned mine = s->census[xcd];
    const unsigned old = __hip_atomic_fetch_add(&s->xcd_arrive[xcd][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    if (old + 1 == mine * gen) {
      __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&s->top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_xcd_active * gen) {
        if (++spins > kSpinLimit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __hip_atomic_store(&s->xcd_release[xcd][0], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&s->xcd_release[xcd][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
        if (++spins > kSpinLimit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
  return ok;
}

__global__ void census_kernel(Sync* s, int only_xcd) {
  if (threadIdx.x == 0) {
    const unsigned x = xcc_id();
    if (only_xcd < 0 || (int)x == only_xcd) atomicAdd(&s->census[x], 1u);
  }
}

// mode 0: flat barrier, 1: hierarchical; only_xcd >= 0: workgroups on other XCDs leave at once
__global__ __launch_bounds__(kThreads) void persistent(float* b0, float* b1, Sync* s, int G, int steps, int mode, int only_xcd,
                                                       int* slot_counter) {
  __shared__ int my_slot;
  const unsigned xcd = xcc_id();
  if (only_xcd >= 0 && (int)xcd != only_xcd) return;
  // logical workgroup number among the participants (dispatch order)
  if (threadIdx.x == 0) my_slot = atomicAdd(slot_counter, 1);
  __syncthreads();
  const int wg = my_slot;
  if (wg >= G) return;                      // more landed on the XCD than asked for
  unsigned n_active = 0;
  for (int x = 0; x < 8; ++x) n_active += s->census[x] > 0;
  unsigned gen = 0;
  int bad = 0;
  bool ok = true;
  for (int step = 0; step < steps && ok; ++step) {
    if (step > 0) bad |= consume(b1, wg, G, 37, 2 * step - 1);
    produce(b0, wg, 2 * step);
    ++gen;
    ok = mode == 0 ? flat_barrier(s, (unsigned)G * gen) : xcd_barrier(s, gen, xcd, n_active);
    if (!ok) break;
    bad |= consume(b0, wg, G, 37, 2 * step);
    produce(b1, wg, 2 * step + 1);
    ++gen;
    ok = mode == 0 ? flat_barrier(s, (unsigned)G * gen) : xcd_barrier(s, gen, xcd, n_active);
  }
  if ((bad || !ok) && threadIdx.x == 0) s->error = bad ? 1 : 2;
}

static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 104;
  const int steps = argc > 2 ? atoi(argv[2]) : 200;
  const int repeats = argc > 3 ? atoi(argv[3]) : 20;
  float *b0, *b1;
  Sync* s;
  int *err, *slot;
  CHECK(hipMalloc(&b0, (size_t)G * kFloatsPerWG * 4));
  CHECK(hipMalloc(&b1, (size_t)G * kFloatsPerWG * 4));
  CHECK(hipMalloc(&s, sizeof(Sync)));
  CHECK(hipMalloc(&err, 4));
  CHECK(hipMalloc(&slot, 4));
  CHECK(hipMemset(err, 0, 4));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));

  // ---- launches: a graph of `steps` x (phase_a, phase_b)
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < steps; ++i) {
    hipLaunchKernelGGL(phase_a, dim3(G), dim3(kThreads), 0, st, b0, b1, G, i, err);
    hipLaunchKernelGGL(phase_b, dim3(G), dim3(kThreads), 0, st, b0, b1, G, i, err);
  }
  CHECK(hipStreamEndCapture(st, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  std::vector<double> t;
  for (int r = 0; r < repeats + 2; ++r) {
    CHECK(hipEventRecord(e0, st));
    CHECK(hipGraphLaunch(exec, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) t.push_back(ms * 1e3 / steps);
  }
  int herr = 0;
  CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("workgroups %d, %d steps of two dependent phases, median of %d runs\n", G, steps, repeats);
  printf("  launches (hipGraph, 2 kernels/step)      %7.2f us/step  (data check %s)\n", median(t), herr ? "FAILED" : "ok");

  // ---- persistent variants
  struct Variant { const char* name; int mode; bool one_xcd; };
  const Variant variants[] = {{"persistent, flat counter barrier     ", 0, false},
                              {"persistent, per-XCD + top barrier    ", 1, false},
                              {"persistent on ONE XCD, flat barrier  ", 0, true}};
  for (const Variant& v : variants) {
    const int grid = v.one_xcd ? G * 8 : G;
    if (v.one_xcd && G > 64) {
      printf("  %s   skipped: %d workgroups of 256 threads do not fit one XCD's 32 CUs x 8 at once\n", v.name, G);
      continue;
    }
    t.clear();
    int result = 0;
    for (int r = 0; r < repeats + 2; ++r) {
      CHECK(hipMemsetAsync(s, 0, sizeof(Sync), st));
      CHECK(hipMemsetAsync(slot, 0, 4, st));
      hipLaunchKernelGGL(census_kernel, dim3(grid), dim3(kThreads), 0, st, s, v.one_xcd ? 0 : -1);
      if (v.one_xcd) {
        // the census may count more than G workgroups on the XCD: the kernel admits the first G, so the barrier counts G
        CHECK(hipStreamSynchronize(st));
        Sync h;
        CHECK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
        if ((int)h.census[0] < G) { result = 3; break; }
        h.census[0] = G;
        CHECK(hipMemcpy(s, &h, sizeof(Sync), hipMemcpyHostToDevice));
      }
      CHECK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent, dim3(grid), dim3(kThreads), 0, st, b0, b1, s, G, steps, v.mode, v.one_xcd ? 0 : -1, slot);
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      Sync h;
      CHECK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
      if (h.error) { result = h.error; break; }
      if (r >= 2) t.push_back(ms * 1e3 / steps);
    }
    if (result)
      printf("  %s   FAILED (%s)\n", v.name, result == 1 ? "stale data" : result == 2 ? "barrier timed out" : "too few workgroups on the XCD");
    else
      printf("  %s %7.2f us/step = %5.2f us per barrier + phase\n", v.name, median(t), median(t) / 2);
  }
  return 0;
}
