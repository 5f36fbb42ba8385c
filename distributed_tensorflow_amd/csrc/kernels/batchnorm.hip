// Training-mode batch normalisation over channels-last bf16 activations, fused with the ReLU and the residual
// addition that follow it in a ResNet bottleneck:
//
//     y = act(gamma * (x - mean) * rstd + beta [+ residual])            x, y, residual: [M, C] bf16, M = N*H*W
//
// As three operators (batch norm, add, ReLU) a layer costs 13 tensor-sized memory passes per training step without
// a residual and 16 with one; here 8 and 10:
//   forward   stats (read x) -> finalize ([C]) -> normalize (read x [, residual], write y [, mask bits])
//   backward  reduce (read dy, x [, mask bits]; write the masked dy when a residual branch needs it) -> finalize ([C])
//             -> dx (read dy, x; write dx)
// The ReLU mask of the backward pass is recomputed from x (no residual) or read as one bit per element that the
// forward pass stored (residual: 1/16 of a tensor pass instead of reading y again).
//
// Layout: a thread owns 8 consecutive channels (one 16-byte load per row) and walks rows with a grid stride;
// C/8 threads cover a row, so a 256-thread block takes 256/(C/8) rows per iteration and every wave reads whole
// contiguous rows.  Column sums go block -> LDS -> one fp32 partial per block and channel; the finalize kernels
// combine the partials in double precision, in a fixed order (results are reproducible run to run).
// Supported: C = 8 * 2^k (every ResNet width from 64 to 2048); blocks take 512 channels at most, wider layers
// are walked by the grid's y dimension.
#include "dtf_kernels.h"

namespace {

// Tuning knobs, measured on the ResNet-50 shapes at batch 256 (tools/bn_bench.py, forward + backward of one layer):
// 512 blocks x unroll 2 with non-temporal stores and non-temporal last-use loads 556 / 788 / 393 us on
// [64, 112x112] / [256, 56x56] + residual / [512, 28x28] + residual; 1024 blocks x unroll 4 with plain accesses 704 /
// 991 / 504.  Fewer, longer-running blocks with fewer loads in flight reach more of the HBM bandwidth (5.9 instead
// of 4.7 TB/s), more blocks (2048) or deeper unrolling (8) less.
#ifndef DTF_BN_MAX_BLOCKS
#define DTF_BN_MAX_BLOCKS 512
#endif
#ifndef DTF_BN_UNROLL
#define DTF_BN_UNROLL 2
#endif
#ifndef DTF_BN_NT_STORE
#define DTF_BN_NT_STORE 1          // outputs are streamed: none is read again before it has left the caches
#endif
#ifndef DTF_BN_NT_LOAD
#define DTF_BN_NT_LOAD 1           // 1: loads that are a tensor's last use in the step; 2: every load
#endif
constexpr int kBnThreads = 256;
constexpr int kBnMaxBlocks = DTF_BN_MAX_BLOCKS;
constexpr int kUnroll = DTF_BN_UNROLL;
constexpr int kChunk = 512;     // channels per block (grid y walks wider layers): keeps the per-block partials small

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
__device__ __forceinline__ void unpack8(const u32x4 q, float* v) {
  v[0] = bf_lo(q.x); v[1] = bf_hi(q.x); v[2] = bf_lo(q.y); v[3] = bf_hi(q.y);
  v[4] = bf_lo(q.z); v[5] = bf_hi(q.z); v[6] = bf_lo(q.w); v[7] = bf_hi(q.w);
}
__device__ __forceinline__ u32x4 pack8(const float* v) {
  u32x4 q;
  q.x = pack_bf(v[0], v[1]); q.y = pack_bf(v[2], v[3]); q.z = pack_bf(v[4], v[5]); q.w = pack_bf(v[6], v[7]);
  return q;
}
__device__ __forceinline__ u32x4 ld16(const unsigned short* p, size_t elem) {
#if DTF_BN_NT_LOAD == 2
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + elem));
#else
  return *reinterpret_cast<const u32x4*>(p + elem);
#endif
}
// last use of a tensor in the step
__device__ __forceinline__ u32x4 ld16_last(const unsigned short* p, size_t elem) {
#if DTF_BN_NT_LOAD >= 1
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + elem));
#else
  return *reinterpret_cast<const u32x4*>(p + elem);
#endif
}
__device__ __forceinline__ void st16(unsigned short* p, size_t elem, u32x4 q) {
#if DTF_BN_NT_STORE
  __builtin_nontemporal_store(q, reinterpret_cast<u32x4*>(p + elem));
#else
  *reinterpret_cast<u32x4*>(p + elem) = q;
#endif
}

struct Walk {          // which rows and channels a thread visits
  int c0;              // first of its 8 channels
  int64_t row, step;   // first row, rows between visits
};
__device__ __forceinline__ int chunk_width(int C) { return C < kChunk ? C : kChunk; }
__device__ __forceinline__ Walk make_walk(int C) {
  const int tpr = chunk_width(C) >> 3, rpi = kBnThreads / tpr;
  Walk w;
  w.c0 = (int)blockIdx.y * chunk_width(C) + (((int)threadIdx.x % tpr) << 3);
  w.row = (int64_t)blockIdx.x * rpi + (int)threadIdx.x / tpr;
  w.step = (int64_t)gridDim.x * rpi;
  return w;
}

// Sums two 8-channel vectors per thread over the threads of the block that own the same channels and stores
// one partial per block: part[(which * C + c) * row_blocks + block] (a channel's partials are contiguous: the
// finalize kernels read them with one wave per channel).
__device__ __forceinline__ void block_column_sums(const float* s0, const float* s1, int C, float* __restrict__ part) {
  __shared__ float red[2][kBnThreads][9];      // 9: threads of one channel group are 1 row apart -> spread the banks
  const int cw = chunk_width(C), tpr = cw >> 3, rpi = kBnThreads / tpr;
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][threadIdx.x][j] = s0[j]; red[1][threadIdx.x][j] = s1[j]; }
  __syncthreads();
  // 2 * cw sums of rpi terms each, dealt to the 256 threads: (which, channel) pairs
  for (int o = threadIdx.x; o < 2 * cw; o += kBnThreads) {
    const int which = o / cw, c = o - which * cw;
    const int t0 = c >> 3, j = c & 7;
    float a = 0.f;
    for (int r = 0; r < rpi; ++r) a += red[which][t0 + r * tpr][j];
    part[((size_t)which * C + (size_t)blockIdx.y * cw + c) * gridDim.x + blockIdx.x] = a;
  }
}

// ---- forward ----
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const unsigned short* __restrict__ x, int64_t M, int C,
                                                               float* __restrict__ part) {
  const Walk w = make_walk(C);
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
  for (int64_t r = w.row; r < M; r += w.step * kUnroll) {
    u32x4 q[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      q[u] = rr < M ? ld16(x, (size_t)rr * C + w.c0) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      float v[8];
      unpack8(q[u], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
    }
  }
  block_column_sums(s, ss, C, part);
}

// save: [4][C] = mean, rstd, a = gamma * rstd, b = beta - mean * a
__global__ __launch_bounds__(kBnThreads) void bn_finalize_fwd_kernel(const float* __restrict__ part, int nblk, int64_t M, int C,
                                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                      float* __restrict__ save, float momentum, float eps) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;    // one wave per channel
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int g = lane; g < nblk; g += 64) {
    s += (double)part[(size_t)c * nblk + g];
    ss += (double)part[((size_t)C + c) * nblk + g];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  if (lane == 0) {
    const double mean = s / (double)M;
    double var = ss / (double)M - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float a = gamma[c] * rstd;
    save[c] = (float)mean;
    save[C + c] = rstd;
    save[2 * C + c] = a;
    save[3 * C + c] = beta[c] - (float)mean * a;
    if (running_mean) {
      const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// bits (residual + ReLU layers): one byte per thread and row, bit j = (output of channel c0 + j is positive): the
// backward pass reads these M * C / 8 bytes instead of the 2 * M * C bytes of y for its ReLU mask.
template <bool RES, bool RELU>
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                                               unsigned short* __restrict__ y, unsigned char* __restrict__ bits,
                                                               const float* __restrict__ save, int64_t M, int C) {
  const Walk w = make_walk(C);
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = save[2 * C + w.c0 + j]; b[j] = save[3 * C + w.c0 + j]; }
  for (int64_t r = w.row; r < M; r += w.step * kUnroll) {
    u32x4 q[kUnroll], qr[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      const bool ok = rr < M;
      q[u] = ok ? ld16(x, (size_t)rr * C + w.c0) : u32x4{0u, 0u, 0u, 0u};
      if (RES) qr[u] = ok ? ld16(res, (size_t)rr * C + w.c0) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      if (rr >= M) continue;
      float v[8], rv[8];
      unpack8(q[u], v);
      if (RES) unpack8(qr[u], rv);
      unsigned m = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float o = v[j] * a[j] + b[j];
        if (RES) o += rv[j];
        if (RELU) o = o > 0.f ? o : 0.f;
        v[j] = o;
      }
      const u32x4 packed = pack8(v);
      if (RES && RELU) {
        // the mask of the backward pass is "the bf16 output is positive": taken from the rounded values
        float rnd[8];
        unpack8(packed, rnd);
#pragma unroll
        for (int j = 0; j < 8; ++j) m |= (rnd[j] > 0.f ? 1u : 0u) << j;
        if (bits) bits[(size_t)rr * (C >> 3) + (w.c0 >> 3)] = (unsigned char)m;
      }
      st16(y, (size_t)rr * C + w.c0, packed);
    }
  }
}

// ---- backward ----
// MASK: 0 = no activation, 1 = ReLU without residual (sign of a*x + b), 2 = ReLU after a residual (the forward
// pass's mask bits; the masked dy is written to dres: it is the gradient of the residual branch and the dx kernel's input).
// Partials: sum(dym) and sum(dym * (x - mean)).
// SUM2 (with MASK 2 only): the output went to two consumers (the next block's main branch and its skip connection) and
// their gradients arrive separately; dy = dy + dy2 is formed here in fp32 instead of by an addition kernel of its own
// (three tensor passes) in front of this one.
template <int MASK, bool SUM2>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_reduce_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ dy2,
                                                                    const unsigned short* __restrict__ x,
                                                                    const unsigned char* __restrict__ bits, unsigned short* __restrict__ dres,
                                                                    const float* __restrict__ save, int64_t M, int C,
                                                                    float* __restrict__ part) {
  const Walk w = make_walk(C);
  float mean[8], a[8], b[8], s[8], sx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = save[w.c0 + j];
    if (MASK == 1) { a[j] = save[2 * C + w.c0 + j]; b[j] = save[3 * C + w.c0 + j]; }
    s[j] = sx[j] = 0.f;
  }
  for (int64_t r = w.row; r < M; r += w.step * kUnroll) {
    u32x4 qd[kUnroll], qd2[kUnroll], qx[kUnroll];
    unsigned mb[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      const bool ok = rr < M;
      const size_t e = (size_t)rr * C + w.c0;
      qd[u] = ok ? (MASK == 2 ? ld16_last(dy, e) : ld16(dy, e)) : u32x4{0u, 0u, 0u, 0u};   // (MASK 2: dx reads dres instead)
      if (SUM2) qd2[u] = ok ? ld16_last(dy2, e) : u32x4{0u, 0u, 0u, 0u};
      qx[u] = ok ? ld16(x, e) : u32x4{0u, 0u, 0u, 0u};
      if (MASK == 2) mb[u] = ok ? (unsigned)bits[(size_t)rr * (C >> 3) + (w.c0 >> 3)] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      if (rr >= M) continue;
      float d[8], d2[8], xv[8];
      unpack8(qd[u], d);
      if (SUM2) unpack8(qd2[u], d2);
      unpack8(qx[u], xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (SUM2) d[j] += d2[j];
        if (MASK == 1) d[j] = (xv[j] * a[j] + b[j]) > 0.f ? d[j] : 0.f;
        if (MASK == 2) d[j] = ((mb[u] >> j) & 1u) ? d[j] : 0.f;
        s[j] += d[j];
        sx[j] += d[j] * (xv[j] - mean[j]);
      }
      if (MASK == 2) st16(dres, (size_t)rr * C + w.c0, pack8(d));
    }
  }
  block_column_sums(s, sx, C, part);
}

// dgamma = rstd * sum(dym * (x - mean)), dbeta = sum(dym);
// coef: [3][C]: dx = c1 * dym + c2 * (x - mean) + c3
__global__ __launch_bounds__(kBnThreads) void bn_finalize_bwd_kernel(const float* __restrict__ part, int nblk, int64_t M, int C,
                                                                      const float* __restrict__ save, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, float* __restrict__ coef) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;    // one wave per channel
  if (c >= C) return;
  double s = 0.0, sx = 0.0;
  for (int g = lane; g < nblk; g += 64) {
    s += (double)part[(size_t)c * nblk + g];
    sx += (double)part[((size_t)C + c) * nblk + g];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); sx += __shfl_xor(sx, o); }
  if (lane == 0) {
    const double rstd = (double)save[C + c], a = (double)save[2 * C + c];
    dgamma[c] = (float)(sx * rstd);
    dbeta[c] = (float)s;
    coef[c] = (float)a;
    coef[C + c] = (float)(-a * rstd * rstd * sx / (double)M);
    coef[2 * C + c] = (float)(-a * s / (double)M);
  }
}

// MASK as above, except that 2 never occurs here: with a residual the input is the masked dy already.
template <int MASK>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_dx_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                unsigned short* __restrict__ dx, const float* __restrict__ save,
                                                                const float* __restrict__ coef, int64_t M, int C) {
  const Walk w = make_walk(C);
  float mean[8], a[8], b[8], c1[8], c2[8], c3[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = save[w.c0 + j];
    if (MASK == 1) { a[j] = save[2 * C + w.c0 + j]; b[j] = save[3 * C + w.c0 + j]; }
    c1[j] = coef[w.c0 + j]; c2[j] = coef[C + w.c0 + j]; c3[j] = coef[2 * C + w.c0 + j];
  }
  for (int64_t r = w.row; r < M; r += w.step * kUnroll) {
    u32x4 qd[kUnroll], qx[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      const bool ok = rr < M;
      const size_t e = (size_t)rr * C + w.c0;
      qd[u] = ok ? ld16_last(dy, e) : u32x4{0u, 0u, 0u, 0u};
      qx[u] = ok ? ld16_last(x, e) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t rr = r + u * w.step;
      if (rr >= M) continue;
      float d[8], xv[8];
      unpack8(qd[u], d);
      unpack8(qx[u], xv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MASK == 1) d[j] = (xv[j] * a[j] + b[j]) > 0.f ? d[j] : 0.f;
        d[j] = c1[j] * d[j] + c2[j] * (xv[j] - mean[j]) + c3[j];
      }
      st16(dx, (size_t)rr * C + w.c0, pack8(d));
    }
  }
}

// grid: x = row blocks (every thread gets two unrolled passes or more when there are rows enough), y = channel chunks
dim3 bn_grid(int64_t M, int C) {
  const int cw = C < kChunk ? C : kChunk, chunks = C / cw;
  const int rpi = kBnThreads / (cw >> 3);
  const int64_t per = (int64_t)rpi * kUnroll * 2;
  const int64_t want = (M + per - 1) / per, cap = kBnMaxBlocks / chunks;
  return dim3((unsigned)(want < 1 ? 1 : (want > cap ? cap : want)), (unsigned)chunks);
}

bool bn_shape_ok(int64_t M, int C) {
  if (M <= 0 || C < 8 || (C & 7) || C > 65536) return false;
  const int g = C >> 3;
  return (g & (g - 1)) == 0;                 // 8 * 2^k channels: the chunking above divides evenly
}

}  // namespace

// Floats of workspace `part` a call needs at most.
DTF_EXPORT int64_t dtf_bn_workspace_floats(int C) { return (int64_t)2 * kBnMaxBlocks * (C < kChunk ? C : kChunk); }

DTF_EXPORT int dtf_bn_supported(int64_t M, int C) { return bn_shape_ok(M, C) ? 1 : 0; }

// y = act(bn(x) [+ res]); save [4][C] receives mean, rstd and the folded scale / shift for the backward pass;
// bits [M * C / 8] bytes (required with res and relu) the ReLU mask the backward pass reads.
DTF_EXPORT int dtf_bn_fwd(const void* x, const void* res, void* y, void* bits, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float* save, float* part, int64_t M, int C, int relu, float momentum, float eps,
                          hipStream_t s) {
  if (!x || !y || !gamma || !beta || !save || !part) return DTF_ERR_ARG;
  if (res && relu && !bits) return DTF_ERR_ARG;
  if (!bn_shape_ok(M, C)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(x) || !dtf_aligned16(y) || (res && !dtf_aligned16(res))) return DTF_ERR_ALIGN;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DTF_ERR_ARG;
  const dim3 grid = bn_grid(M, C);
  const int nblk = (int)grid.x;
  const auto* xs = static_cast<const unsigned short*>(x);
  const auto* rs = static_cast<const unsigned short*>(res);
  auto* ys = static_cast<unsigned short*>(y);
  auto* bs = static_cast<unsigned char*>(bits);
  hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(kBnThreads), 0, s, xs, M, C, part);
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(dtf_cdiv(C, 4)), dim3(kBnThreads), 0, s, part, nblk, M, C, gamma, beta, running_mean,
                     running_var, save, momentum, eps);
  if (res && relu) hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, dim3(kBnThreads), 0, s, xs, rs, ys, bs, save, M, C);
  else if (res) hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, dim3(kBnThreads), 0, s, xs, rs, ys, bs, save, M, C);
  else if (relu) hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, dim3(kBnThreads), 0, s, xs, rs, ys, bs, save, M, C);
  else hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, dim3(kBnThreads), 0, s, xs, rs, ys, bs, save, M, C);
  return dtf_check_launch();
}

// Backward of the above.  mask: 0 none, 1 ReLU recomputed from x, 2 ReLU from the forward pass's mask bits (then `bits`
// and `dres` are required and dres receives the masked dy = the gradient of the residual input).  coef: [3][C] scratch.
// dy2 (optional, mask 2 only): a second gradient of the same output; the layer's dy is dy + dy2.
DTF_EXPORT int dtf_bn_bwd(const void* dy, const void* dy2, const void* x, const void* bits, void* dx, void* dres, const float* save,
                          float* dgamma, float* dbeta, float* part, float* coef, int64_t M, int C, int mask, hipStream_t s) {
  if (!dy || !x || !dx || !save || !dgamma || !dbeta || !part || !coef) return DTF_ERR_ARG;
  if (mask < 0 || mask > 2 || (mask == 2 && (!bits || !dres)) || (dy2 && mask != 2)) return DTF_ERR_ARG;
  if (dy2 && !dtf_aligned16(dy2)) return DTF_ERR_ALIGN;
  if (!bn_shape_ok(M, C)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(dy) || !dtf_aligned16(x) || !dtf_aligned16(dx) || (mask == 2 && !dtf_aligned16(dres)))
    return DTF_ERR_ALIGN;
  const dim3 grid = bn_grid(M, C);
  const int nblk = (int)grid.x;
  const auto* ds = static_cast<const unsigned short*>(dy);
  const auto* ds2 = static_cast<const unsigned short*>(dy2);
  const auto* xs = static_cast<const unsigned short*>(x);
  const auto* mbits = static_cast<const unsigned char*>(bits);
  auto* dxs = static_cast<unsigned short*>(dx);
  auto* drs = static_cast<unsigned short*>(dres);
  if (mask == 2 && ds2) hipLaunchKernelGGL((bn_bwd_reduce_kernel<2, true>), grid, dim3(kBnThreads), 0, s, ds, ds2, xs, mbits, drs, save, M, C, part);
  else if (mask == 2) hipLaunchKernelGGL((bn_bwd_reduce_kernel<2, false>), grid, dim3(kBnThreads), 0, s, ds, ds2, xs, mbits, drs, save, M, C, part);
  else if (mask == 1) hipLaunchKernelGGL((bn_bwd_reduce_kernel<1, false>), grid, dim3(kBnThreads), 0, s, ds, ds2, xs, mbits, drs, save, M, C, part);
  else hipLaunchKernelGGL((bn_bwd_reduce_kernel<0, false>), grid, dim3(kBnThreads), 0, s, ds, ds2, xs, mbits, drs, save, M, C, part);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(dtf_cdiv(C, 4)), dim3(kBnThreads), 0, s, part, nblk, M, C, save, dgamma, dbeta, coef);
  if (mask == 2) hipLaunchKernelGGL(bn_bwd_dx_kernel<0>, grid, dim3(kBnThreads), 0, s, drs, xs, dxs, save, coef, M, C);
  else if (mask == 1) hipLaunchKernelGGL(bn_bwd_dx_kernel<1>, grid, dim3(kBnThreads), 0, s, ds, xs, dxs, save, coef, M, C);
  else hipLaunchKernelGGL(bn_bwd_dx_kernel<0>, grid, dim3(kBnThreads), 0, s, ds, xs, dxs, save, coef, M, C);
  return dtf_check_launch();
}
