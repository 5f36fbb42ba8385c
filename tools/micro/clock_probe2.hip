// Micro-probe 2: cost of the 16-lane DPP reductions and of one softmax-gradient row (the unit of work the fused
// backward kernel repeats per 16 batch rows), single wave, in shader clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DTF_ROR(v, n) __builtin_amdgcn_update_dpp((v), (v), 0x120 + (n), 0xf, 0xf, false)
__device__ __forceinline__ float row16_sum(float v) { v += DTF_ROR(v, 8); v += DTF_ROR(v, 4); v += DTF_ROR(v, 2); v += DTF_ROR(v, 1); return v; }
__device__ __forceinline__ float row16_max(float v) { v = fmaxf(v, DTF_ROR(v, 8)); v = fmaxf(v, DTF_ROR(v, 4)); v = fmaxf(v, DTF_ROR(v, 2)); v = fmaxf(v, DTF_ROR(v, 1)); return v; }
__device__ __forceinline__ float row_math(float z, float lab, float inv_b) {
  const float zmax = row16_max(z);
  const float e1 = __expf(z - zmax);
  const float s1 = row16_sum(e1);
  const float y = e1 * __builtin_amdgcn_rcpf(s1);
  const float ymax = row16_max(y);
  const float e2 = __expf(y - ymax);
  const float s2 = row16_sum(e2);
  const float pp = e2 * __builtin_amdgcn_rcpf(s2);
  const float dy = (pp - lab) * inv_b;
  const float dot = row16_sum(dy * y);
  return (dy - dot) * y;
}
__global__ void probe(unsigned long long* out, float* data, int n) {
  float v = data[threadIdx.x], lab = data[64 + threadIdx.x];
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v = row16_sum(v) * 0.0625f + 1.0f;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float w = v;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) w = row_math(w, lab, 0.01f) + 1.0f;      // dependent rows
  }
  unsigned long long c2 = __builtin_amdgcn_s_memtime();
  float a = w, b = w + 1.f, c = w + 2.f, d = w + 3.f;
  for (int i = 0; i < n; ++i) {                                           // four independent rows interleaved
    a = row_math(a, lab, 0.01f) + 1.0f; b = row_math(b, lab, 0.01f) + 1.0f; c = row_math(c, lab, 0.01f) + 1.0f; d = row_math(d, lab, 0.01f) + 1.0f;
  }
  unsigned long long c3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = c2 - c1; out[2] = c3 - c2; }
  data[128 + threadIdx.x] = a + b + c + d;
}
int main() {
  unsigned long long* out; float* data;
  hipMalloc(&out, 64); hipMalloc(&data, 4 * 1024); hipMemset(data, 0, 4 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    const int n = 100;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, data, n);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("row16_sum (+fma) %.1f clk; dependent softmax-gradient row %.1f clk; four interleaved rows %.1f clk per row\n",
           (double)h[0] / (8 * n), (double)h[1] / (4 * n), (double)h[2] / (4 * n));
  }
  return 0;
}
