// Device-side helpers shared by the latency kernels (gfx950 only): fp32 MFMA shorthand, raw buffer
// loads, 16-lane DPP reductions and the step tail.
#pragma once
#include "dtf_kernels.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- raw buffer access: byte offsets, out-of-range -> 0 -------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static constexpr unsigned kOOB = 0xFFFFFFF0u;   // beyond any num_records: the load returns 0
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, unsigned off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ int bldi(rsrc_t r, unsigned off) { return (int)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0); }
__device__ __forceinline__ void bld4(rsrc_t r, unsigned off, float (&v)[4]) {
  auto t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
}

// ---- reductions over a 16-lane row by DPP rotation (every lane ends with the result) ---------------
#define DTF_ROR(v, n) __builtin_amdgcn_update_dpp((v), (v), 0x120 + (n), 0xf, 0xf, false)
__device__ __forceinline__ float row16_sum(float v) {
  v += DTF_ROR(v, 8); v += DTF_ROR(v, 4); v += DTF_ROR(v, 2); v += DTF_ROR(v, 1);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, DTF_ROR(v, 8)); v = fmaxf(v, DTF_ROR(v, 4)); v = fmaxf(v, DTF_ROR(v, 2)); v = fmaxf(v, DTF_ROR(v, 1));
  return v;
}
__device__ __forceinline__ int row16_min(int v) {
  v = min(v, DTF_ROR(v, 8)); v = min(v, DTF_ROR(v, 4)); v = min(v, DTF_ROR(v, 2)); v = min(v, DTF_ROR(v, 1));
  return v;
}

// --------------------------------------------------------------------------------------
// step tail.  Runs in ONE workgroup of exactly 256 threads.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ void dtf_tail_body(const DtfTailArgs& t, int tid, int nthreads, bool init_only) {
  __shared__ float s_loss[256];
  __shared__ int s_cnt[256][2];
  // Every thread reads the counters before thread 0 writes them (barriers below).
  const int gs = t.state[0];
  const int ls = t.state[1];
  if (!init_only) {
    float loss = 0.f;
    int correct = 0, post = 0;
    for (int i = tid; i < t.nparts; i += nthreads) {
      loss += t.loss_part[i];
      correct += t.correct_part[i];
      post += t.post_part[i];
    }
    s_loss[tid] = loss;
    s_cnt[tid][0] = correct;
    s_cnt[tid][1] = post;
    __syncthreads();
    // fixed-shape tree: the logged loss is bitwise reproducible
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        s_loss[tid] += s_loss[tid + s];
        s_cnt[tid][0] += s_cnt[tid + s][0];
        s_cnt[tid][1] += s_cnt[tid + s][1];
      }
      __syncthreads();
    }
    if (tid == 0) {
      if (t.log) {
        int* rec = t.log + 4 * (ls % t.log_cap);
        rec[0] = __float_as_int(s_loss[0] * t.inv_b);
        rec[1] = s_cnt[0][0];
        rec[2] = s_cnt[0][1];
        rec[3] = gs;  // value of global_step before this step's increment
      }
      t.state[0] = gs + t.gs_inc;
      t.state[1] = ls + 1;
    }
  } else {
    __syncthreads();
  }
  if (t.ridx_next) {
    const int next = init_only ? ls : ls + 1;
    for (int i = tid; i < t.B; i += nthreads) {
      const long long p = (long long)next * t.B + i;
      const long long epoch = p / t.n_train;
      const int pos = (int)(p - epoch * t.n_train);
      const int row = t.perm ? t.perm[(size_t)(epoch & 1) * t.n_train + pos] : pos;
      const int prev = (init_only || !t.ridx_cur) ? row : t.ridx_cur[i];
      t.ridx_next[t.B + i] = prev;  // the batch just trained on becomes the accuracy-only rows
      t.ridx_next[i] = row;
    }
  }
}

