// Flat fused shard updates and parameter initialisation.
//
// The parameter-server role of /root/reference/distributed.py (variables on
// /job:ps, ApplyGradientDescent executed there, :60-64, :89, :102) becomes: each
// rank owns a contiguous slice of one flat fp32 parameter buffer in HBM and
// applies the optimizer to that slice with one launch.  These kernels are pure
// streaming (16 B per lane, grid-stride, <= 2048 workgroups), i.e. HBM-bound.
#include "dtf_kernels.h"

static inline int stream_grid(int64_t nvec, int block) {
  int64_t g = (nvec + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  return (int)g;
}

// theta -= lr * scale * g        (scale = 1/R turns a reduce-scatter sum into the SyncReplicas mean)
__global__ __launch_bounds__(256) void sgd_apply_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n,
                                                        float step) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    pv.x -= step * gv.x; pv.y -= step * gv.y; pv.z -= step * gv.z; pv.w -= step * gv.w;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    p[i] -= step * g[i];
  }
}

DTF_EXPORT int dtf_sgd_apply(float* param, const float* grad, int64_t n, float lr, float grad_scale, hipStream_t s) {
  if (!param || !grad) return DTF_ERR_ARG;
  if (n <= 0) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(param) || !dtf_aligned16(grad)) return DTF_ERR_ALIGN;
  hipLaunchKernelGGL(sgd_apply_kernel, dim3(stream_grid(n >> 2, 256)), dim3(256), 0, s, param, grad, n, lr * grad_scale);
  return dtf_check_launch();
}

// Adam with bias correction; the step number is read from device memory so the
// launch can sit inside a replayed hipGraph.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps_t) {
  dtf_adam_one(p, g, m, v, lr_t, b1, b2, eps_t);
}

__global__ __launch_bounds__(256) void adam_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                         float b1, float b2, float eps, float gscale,
                                                         const int* __restrict__ step_ptr, int step_offset) {
  // t counts applied updates starting at 1
  const int t = (step_ptr ? *step_ptr : 0) + step_offset;
  const float bc1 = 1.f - powf(b1, (float)t);
  const float bc2 = 1.f - powf(b2, (float)t);
  // TF-style folding: lr_t = lr * sqrt(1-b2^t)/(1-b1^t), epsilon applied to sqrt(v) un-corrected is
  // "epsilon hat"; we keep the textbook form by scaling eps with sqrt(bc2).
  const float sq = sqrtf(bc2);
  const float lr_t = lr * sq / bc1;
  const float eps_t = eps * sq;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // two independent 16-byte positions per thread and iteration (eight loads in flight before the first use);
  // everything is touched exactly once per launch, hence the non-temporal accesses
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += 2 * stride) {
    const int64_t j = i + stride < nvec ? i + stride : i;      // odd tail: position i twice, second result unused
    f32x4 pv = __builtin_nontemporal_load(p4 + i), gv = __builtin_nontemporal_load(g4 + i);
    f32x4 mv = __builtin_nontemporal_load(m4 + i), vv = __builtin_nontemporal_load(v4 + i);
    f32x4 pw = __builtin_nontemporal_load(p4 + j), gw = __builtin_nontemporal_load(g4 + j);
    f32x4 mw = __builtin_nontemporal_load(m4 + j), vw = __builtin_nontemporal_load(v4 + j);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = pv[k], b = mv[k], c = vv[k];
      adam_one(a, gv[k] * gscale, b, c, lr_t, b1, b2, eps_t);
      pv[k] = a; mv[k] = b; vv[k] = c;
    }
    __builtin_nontemporal_store(pv, p4 + i);
    __builtin_nontemporal_store(mv, m4 + i);
    __builtin_nontemporal_store(vv, v4 + i);
    if (j != i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a = pw[k], b = mw[k], c = vw[k];
        adam_one(a, gw[k] * gscale, b, c, lr_t, b1, b2, eps_t);
        pw[k] = a; mw[k] = b; vw[k] = c;
      }
      __builtin_nontemporal_store(pw, p4 + j);
      __builtin_nontemporal_store(mw, m4 + j);
      __builtin_nontemporal_store(vw, v4 + j);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    adam_one(p[i], g[i] * gscale, m[i], v[i], lr_t, b1, b2, eps_t);
  }
}

DTF_EXPORT int dtf_adam_apply(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float grad_scale, const int* step_ptr, int step_offset,
                              hipStream_t s) {
  if (!param || !grad || !m || !v) return DTF_ERR_ARG;
  if (n <= 0) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(param) || !dtf_aligned16(grad) || !dtf_aligned16(m) || !dtf_aligned16(v)) return DTF_ERR_ALIGN;
  hipLaunchKernelGGL(adam_apply_kernel, dim3(stream_grid(n >> 2, 256)), dim3(256), 0, s, param, grad, m, v, n, lr, beta1,
                     beta2, eps, grad_scale, step_ptr, step_offset);
  return dtf_check_launch();
}

// The optimizer over up to DTF_MAX_OPT_RANGES ranges [lo, hi) of one flat buffer in ONE launch (the pieces the
// weight-gradient epilogues leave over: biases, small layers): blockIdx.y = range.  Same arithmetic as above.
__global__ __launch_bounds__(256) void opt_ranges_kernel(DtfOptRangesArgs a) {
  const int64_t lo = a.lo[blockIdx.y], n4 = (a.hi[blockIdx.y] - lo) >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(a.p + lo);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(a.g + lo);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (!a.m) {
    const float step = a.lr * a.gscale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) p4[i] = p4[i] - step * g4[i];
    return;
  }
  const int t = (a.step_ptr ? *a.step_ptr : 0) + a.step_offset;
  const float bc1 = 1.f - powf(a.b1, (float)t), bc2 = 1.f - powf(a.b2, (float)t);
  const float sq = sqrtf(bc2);
  const float lr_t = a.lr * sq / bc1, eps_t = a.eps * sq;
  f32x4* m4 = reinterpret_cast<f32x4*>(a.m + lo);
  f32x4* v4 = reinterpret_cast<f32x4*>(a.v + lo);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = p4[i], gv = g4[i], mv = m4[i], vv = v4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float x = pv[k], y = mv[k], z = vv[k];
      adam_one(x, gv[k] * a.gscale, y, z, lr_t, a.b1, a.b2, eps_t);
      pv[k] = x; mv[k] = y; vv[k] = z;
    }
    p4[i] = pv; m4[i] = mv; v4[i] = vv;
  }
}

DTF_EXPORT int dtf_opt_ranges(const DtfOptRangesArgs* a, hipStream_t s) {
  if (!a || !a->p || !a->g || (a->m == nullptr) != (a->v == nullptr)) return DTF_ERR_ARG;
  if (a->nranges <= 0 || a->nranges > DTF_MAX_OPT_RANGES) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->p) || !dtf_aligned16(a->g) || (a->m && (!dtf_aligned16(a->m) || !dtf_aligned16(a->v)))) return DTF_ERR_ALIGN;
  int64_t longest = 0;
  for (int i = 0; i < a->nranges; ++i) {
    if (a->lo[i] < 0 || a->hi[i] <= a->lo[i] || (a->lo[i] & 3) || (a->hi[i] & 3)) return DTF_ERR_SHAPE;
    if (a->hi[i] - a->lo[i] > longest) longest = a->hi[i] - a->lo[i];
  }
  hipLaunchKernelGGL(opt_ranges_kernel, dim3(stream_grid(longest >> 2, 256), a->nranges), dim3(256), 0, s, *a);
  return dtf_check_launch();
}

// dst = alpha * dst + beta * src   (gradient accumulation over several rounds, R > num_workers)
__global__ __launch_bounds__(256) void accumulate_kernel(float* __restrict__ d, const float* __restrict__ s, int64_t n,
                                                         float alpha, float beta) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float4 dv = reinterpret_cast<float4*>(d)[i];
    const float4 sv = reinterpret_cast<const float4*>(s)[i];
    dv.x = alpha * dv.x + beta * sv.x; dv.y = alpha * dv.y + beta * sv.y;
    dv.z = alpha * dv.z + beta * sv.z; dv.w = alpha * dv.w + beta * sv.w;
    reinterpret_cast<float4*>(d)[i] = dv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    d[i] = alpha * d[i] + beta * s[i];
  }
}

DTF_EXPORT int dtf_accumulate(float* dst, const float* src, int64_t n, float alpha, float beta, hipStream_t s) {
  if (!dst || !src) return DTF_ERR_ARG;
  if (n <= 0) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(dst) || !dtf_aligned16(src)) return DTF_ERR_ALIGN;
  hipLaunchKernelGGL(accumulate_kernel, dim3(stream_grid(n >> 2, 256)), dim3(256), 0, s, dst, src, n, alpha, beta);
  return dtf_check_launch();
}

// --------------------------------------------------------------------------------------
// Upload by kernel: dst (device) = src (page-locked HOST memory, mapped into the device's address space), 4-byte
// words, read over the host link by the shader cores.  The written data passes through the same caches as any kernel
// output, so the kernels that read it next are ordered and coherent with it by the ordinary stream rules, and an
// epoch's permutation can be uploaded in the middle of a run without draining the stream (the copy is one more
// kernel in stream order; the staging buffer is guarded by an event).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upload_words_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

DTF_EXPORT int dtf_upload_words(void* dst, const void* host_src, int64_t n_words, hipStream_t s) {
  if (!dst || !host_src) return DTF_ERR_ARG;
  if (n_words <= 0) return DTF_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(host_src)) & 3u) return DTF_ERR_ALIGN;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, host_src) != hipSuccess || at.type != hipMemoryTypeHost || at.devicePointer == nullptr) {
    (void)hipGetLastError();
    return DTF_ERR_ARG;                              // not page-locked, mapped host memory
  }
  hipLaunchKernelGGL(upload_words_kernel, dim3(stream_grid(n_words, 256)), dim3(256), 0, s, static_cast<unsigned*>(dst),
                     static_cast<const unsigned*>(at.devicePointer), n_words);
  return dtf_check_launch();
}

// --------------------------------------------------------------------------------------
// truncated normal initialiser (tf.truncated_normal, distributed.py:67-72): N(0, stddev)
// re-sampled until |v| <= 2 stddev.  Counter-based Philox4x32-10, one counter block per
// (element, attempt) so the result does not depend on the launch geometry.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(256) void truncated_normal_kernel(float* __restrict__ out, int64_t n, float stddev,
                                                               uint64_t seed, uint64_t stream_id) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float val = 0.f;
    bool done = false;
    for (uint32_t attempt = 0; attempt < 64 && !done; ++attempt) {
      uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), attempt, (uint32_t)stream_id};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
      // two Box-Muller pairs per counter block
      const float u0 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u1 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u3 = ((float)(c[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
      const float z[4] = {r0 * cosf(6.28318530718f * u1), r0 * sinf(6.28318530718f * u1),
                          r1 * cosf(6.28318530718f * u3), r1 * sinf(6.28318530718f * u3)};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!done && fabsf(z[j]) <= 2.f) { val = z[j]; done = true; }
    }
    out[i] = val * stddev;
  }
}

DTF_EXPORT int dtf_truncated_normal(float* out, int64_t n, float stddev, uint64_t seed, uint64_t stream_id, hipStream_t s) {
  if (!out) return DTF_ERR_ARG;
  if (n <= 0) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(truncated_normal_kernel, dim3(stream_grid(n, 256)), dim3(256), 0, s, out, n, stddev, seed, stream_id);
  return dtf_check_launch();
}

DTF_EXPORT int dtf_run_schedule(const DtfScheduleEntry* entries, int n, hipStream_t s) {
  if (!entries || n < 0) return DTF_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    int rc;
    switch (entries[i].op) {
      case DTF_OP_SMALL_GEMM: rc = dtf_small_gemm(static_cast<const DtfSmallGemmArgs*>(entries[i].args), s); break;
      case DTF_OP_HEAD: rc = dtf_head(static_cast<const DtfHeadArgs*>(entries[i].args), s); break;
      case DTF_OP_WGRAD: rc = dtf_wgrad_grouped(static_cast<const DtfWgradArgs*>(entries[i].args), s); break;
      case DTF_OP_TAIL: rc = dtf_step_tail(static_cast<const DtfTailArgs*>(entries[i].args), entries[i].flag, s); break;
      case DTF_OP_GEMM_TILED: rc = dtf_gemm_tiled(static_cast<const DtfGemmArgs*>(entries[i].args), s); break;
      case DTF_OP_UPDATE: rc = dtf_run_update(static_cast<const DtfUpdateArgs*>(entries[i].args), s); break;
      case DTF_OP_MLP2_FWD: rc = dtf_mlp2_fwd(static_cast<const DtfMlp2Args*>(entries[i].args), s); break;
      case DTF_OP_MLP2_BWD: rc = dtf_mlp2_bwd(static_cast<const DtfMlp2Args*>(entries[i].args), s); break;
      case DTF_OP_P2P_UPDATE: rc = dtf_p2p_update(static_cast<const DtfP2PUpdateArgs*>(entries[i].args), s); break;
      case DTF_OP_ASYNC_PUSH_PULL: rc = dtf_async_push_pull(static_cast<const DtfAsyncPushPullArgs*>(entries[i].args), s); break;
      case DTF_OP_P2P_SYNC_GENERAL: rc = dtf_p2p_sync_general(static_cast<const DtfP2PGeneralArgs*>(entries[i].args), s); break;
      case DTF_OP_SOFTMAX_XENT: rc = dtf_softmax_xent(static_cast<const DtfXentArgs*>(entries[i].args), s); break;
      default: rc = DTF_ERR_ARG;
    }
    if (rc != DTF_OK) return (i << 8) | rc;
  }
  return DTF_OK;
}

// Several steps in one call: step i runs the schedule of slot (first_slot + i) % nslots.  Launching a short run
// this way starts the device a few microseconds after the call (a graph replay of the same steps takes 10-16 us
// to start) and the host stays ahead of the device as long as a kernel runs longer than its launch call.
DTF_EXPORT int dtf_run_schedule_steps(const DtfScheduleEntry* const* per_slot, const int* counts, int nslots, int first_slot,
                                      int n_steps, hipStream_t s) {
  if (!per_slot || !counts || nslots <= 0 || first_slot < 0 || first_slot >= nslots || n_steps < 0) return DTF_ERR_ARG;
  int q = first_slot;
  for (int i = 0; i < n_steps; ++i) {
    const int rc = dtf_run_schedule(per_slot[q], counts[q], s);
    if (rc != DTF_OK) return rc;
    q = q + 1 == nslots ? 0 : q + 1;
  }
  return DTF_OK;
}

DTF_EXPORT int dtf_kernels_abi_version() { return 1; }
