// Max pooling over channels-last bf16 activations (the 3x3 / stride 2 pooling behind the first convolution of a
// ResNet: [256, 112, 112, 64] -> [256, 56, 56, 64] at batch 256).
//
// The forward pass stores, next to the maxima, the position of each maximum inside its window as ONE BYTE (kh * k + kw,
// first maximum in scan order); the backward pass is a gather: every input element looks at the windows that
// contain it (at most ceil(k/s)^2, four here) and adds the output gradients of those whose stored position is its own.
// No atomics, no recomputation, every dx element written once; per step the backward pass moves position bytes
// (1/8 of dx) + dy (1/4) + dx, where a kernel working from 8-byte element indices reads 4x the bytes of dy for them.
//
// A thread owns 8 consecutive channels (one 16-byte access); consecutive threads walk channels, then width.
// Measured on [256, 64, 112, 112] (tools/maxpool_bench.py): forward 189 us (F.max_pool2d 270), forward + backward
// 623 us (1146).  A forward variant that issues all nine window loads before the first comparison (clamped addresses,
// masked positions) was slower, 275 us: one load at a time per thread, many threads, is the better pattern here.
#include "dtf_kernels.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mp_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float mp_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ void mp_unpack8(const u32x4 q, float* v) {
  v[0] = mp_lo(q.x); v[1] = mp_hi(q.x); v[2] = mp_lo(q.y); v[3] = mp_hi(q.y);
  v[4] = mp_lo(q.z); v[5] = mp_hi(q.z); v[6] = mp_lo(q.w); v[7] = mp_hi(q.w);
}
__device__ __forceinline__ unsigned mp_pack(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}

struct PoolDims {
  int N, H, W, C, OH, OW, k, s, p;
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                           unsigned char* __restrict__ pos, PoolDims d) {
  // (element counts fit 31 bits, checked on the host: the index arithmetic is 32-bit, a 64-bit division costs ~10x)
  const unsigned c8n = (unsigned)d.C >> 3;
  const unsigned total = (unsigned)d.N * d.OH * d.OW * c8n;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t % c8n);
  unsigned r = t / c8n;
  const int ow = (int)(r % (unsigned)d.OW);
  r /= (unsigned)d.OW;
  const int oh = (int)(r % (unsigned)d.OH);
  const int n = (int)(r / (unsigned)d.OH);
  float best[8];
  unsigned bp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { best[j] = -__builtin_inff(); bp[j] = 0xffu; }     // 0xff: nothing seen yet
  for (int kh = 0; kh < d.k; ++kh) {
    const int ih = oh * d.s - d.p + kh;
    if (ih < 0 || ih >= d.H) continue;
    for (int kw = 0; kw < d.k; ++kw) {
      const int iw = ow * d.s - d.p + kw;
      if (iw < 0 || iw >= d.W) continue;
      const u32x4 q = *reinterpret_cast<const u32x4*>(x + (((size_t)n * d.H + ih) * d.W + iw) * d.C + (size_t)c8 * 8);
      float v[8];
      mp_unpack8(q, v);
      const unsigned here = (unsigned)(kh * d.k + kw);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // first maximum in scan order; a NaN wins and stays
        const bool nan_kept = best[j] != best[j];
        const bool take = bp[j] == 0xffu || (!nan_kept && (v[j] > best[j] || v[j] != v[j]));
        best[j] = take ? v[j] : best[j];
        bp[j] = take ? here : bp[j];
      }
    }
  }
  u32x4 o;
  o.x = mp_pack(best[0], best[1]); o.y = mp_pack(best[2], best[3]); o.z = mp_pack(best[4], best[5]); o.w = mp_pack(best[6], best[7]);
  __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(y + (size_t)t * 8));
  u32x2 pb;
  pb.x = bp[0] | (bp[1] << 8) | (bp[2] << 16) | (bp[3] << 24);
  pb.y = bp[4] | (bp[5] << 8) | (bp[6] << 16) | (bp[7] << 24);
  *reinterpret_cast<u32x2*>(pos + (size_t)t * 8) = pb;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ pos,
                                                           unsigned short* __restrict__ dx, PoolDims d) {
  const unsigned c8n = (unsigned)d.C >> 3;
  const unsigned total = (unsigned)d.N * d.H * d.W * c8n;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int c8 = (int)(t % c8n);
  unsigned r = t / c8n;
  const int iw = (int)(r % (unsigned)d.W);
  r /= (unsigned)d.W;
  const int ih = (int)(r % (unsigned)d.H);
  const int n = (int)(r / (unsigned)d.H);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // windows that contain row ih: oh * s - p <= ih <= oh * s - p + k - 1, i.e. ceil((ih + p - k + 1) / s) <= oh <= (ih + p) / s
  const int nh = ih + d.p - d.k + 1, nw = iw + d.p - d.k + 1;
  const int oh_lo = nh <= 0 ? 0 : (nh + d.s - 1) / d.s, ow_lo = nw <= 0 ? 0 : (nw + d.s - 1) / d.s;
  const int oh_hi = min((ih + d.p) / d.s, d.OH - 1), ow_hi = min((iw + d.p) / d.s, d.OW - 1);
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    const int kh = ih + d.p - oh * d.s;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const int kw = iw + d.p - ow * d.s;
      const unsigned here = (unsigned)(kh * d.k + kw);
      const size_t o = ((((size_t)n * d.OH + oh) * d.OW + ow) * c8n + c8) * 8;
      const u32x2 pb = *reinterpret_cast<const u32x2*>(pos + o);
      const u32x4 q = *reinterpret_cast<const u32x4*>(dy + o);
      float g[8];
      mp_unpack8(q, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned b = ((j < 4 ? pb.x : pb.y) >> (8 * (j & 3))) & 0xffu;
        if (b == here) acc[j] += g[j];
      }
    }
  }
  u32x4 o;
  o.x = mp_pack(acc[0], acc[1]); o.y = mp_pack(acc[2], acc[3]); o.z = mp_pack(acc[4], acc[5]); o.w = mp_pack(acc[6], acc[7]);
  __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(dx + (size_t)t * 8));
}

bool pool_ok(const PoolDims& d) {
  if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.C < 8 || (d.C & 7)) return false;
  if (d.k < 1 || d.k > 15 || d.s < 1 || d.p < 0 || 2 * d.p > d.k) return false;     // positions fit a byte; no window lies in the padding
  if (d.OH != (d.H + 2 * d.p - d.k) / d.s + 1 || d.OW != (d.W + 2 * d.p - d.k) / d.s + 1 || d.OH < 1 || d.OW < 1) return false;
  const int64_t in = (int64_t)d.N * d.H * d.W * (d.C >> 3);      // threads of the backward launch (the larger one)
  return in < 0x7fffff00LL;                                       // 32-bit thread indices in the kernels
}

}  // namespace

// y [N, OH, OW, C] bf16 and pos [N, OH, OW, C] bytes from x [N, H, W, C] bf16 (floor mode, no dilation).
DTF_EXPORT int dtf_maxpool_fwd(const void* x, void* y, void* pos, int N, int H, int W, int C, int OH, int OW, int k, int s, int p,
                               hipStream_t st) {
  if (!x || !y || !pos) return DTF_ERR_ARG;
  const PoolDims d{N, H, W, C, OH, OW, k, s, p};
  if (!pool_ok(d)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(x) || !dtf_aligned16(y) || !dtf_aligned16(pos)) return DTF_ERR_ALIGN;
  const int64_t total = (int64_t)N * OH * OW * (C >> 3);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned short*>(x),
                     static_cast<unsigned short*>(y), static_cast<unsigned char*>(pos), d);
  return dtf_check_launch();
}

// dx [N, H, W, C] bf16 from dy [N, OH, OW, C] bf16 and the positions the forward pass stored.
DTF_EXPORT int dtf_maxpool_bwd(const void* dy, const void* pos, void* dx, int N, int H, int W, int C, int OH, int OW, int k, int s,
                               int p, hipStream_t st) {
  if (!dy || !dx || !pos) return DTF_ERR_ARG;
  const PoolDims d{N, H, W, C, OH, OW, k, s, p};
  if (!pool_ok(d)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(dy) || !dtf_aligned16(dx) || !dtf_aligned16(pos)) return DTF_ERR_ALIGN;
  const int64_t total = (int64_t)N * H * W * (C >> 3);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned short*>(dy),
                     static_cast<const unsigned char*>(pos), static_cast<unsigned short*>(dx), d);
  return dtf_check_launch();
}
