// Micro-probe: shader clock vs the 100 MHz constant clock, latency of dependent VALU / DPP / exp / MFMA chains,
// and of an L2 round trip, inside a short-lived kernel (the regime the latency kernels run in).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ROR(v, n) __builtin_amdgcn_update_dpp((v), (v), 0x120 + (n), 0xf, 0xf, false)
__global__ void probe(unsigned long long* out, float* data, int n) {
  float v = data[threadIdx.x];
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  _Pragma("unroll 16") for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;          // dependent FMA chain
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float w = v;
  _Pragma("unroll 16") for (int i = 0; i < n; ++i) w += ROR(w, 1);                  // dependent DPP add chain
  unsigned long long c2 = __builtin_amdgcn_s_memtime();
  float e = w;
  _Pragma("unroll 16") for (int i = 0; i < n; ++i) e = __expf(e) * 0.25f;            // dependent exp chain
  unsigned long long c3 = __builtin_amdgcn_s_memtime();
  f32x4 acc = {e, e, e, e};
_Pragma("unroll 8")  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[0], 1.0f, acc, 0, 0, 0);   // dependent MFMA chain
  unsigned long long c4 = __builtin_amdgcn_s_memtime();
  f32x4 a0 = acc, a1 = acc;
_Pragma("unroll 8")  for (int i = 0; i < n; ++i) { a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, a1, 0, 0, 0); }
  unsigned long long c5 = __builtin_amdgcn_s_memtime();
  // pointer chase through L2 (data[] holds indices as floats)
  int idx = (int)data[1024 + threadIdx.x] & 1023;
  for (int i = 0; i < 64; ++i) idx = (int)data[1024 + idx * 64 % 65536] & 1023;
  unsigned long long c6 = __builtin_amdgcn_s_memtime(), r6 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0; out[1] = r1 - r0; out[2] = c2 - c1; out[3] = c3 - c2; out[4] = c4 - c3; out[5] = c5 - c4; out[6] = c6 - c5; out[7] = r6 - r0;
  }
  data[threadIdx.x + 70000] = a0[0] + a1[1] + (float)idx;
}
int main() {
  unsigned long long* out; float* data;
  hipMalloc(&out, 64); hipMalloc(&data, 4 * 80000); hipMemset(data, 0, 4 * 80000);
  for (int rep = 0; rep < 4; ++rep) {
    const int n = rep < 2 ? 2000 : 200;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, data, n);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("n=%d: fma chain %.1f clk/op; sclk/100MHz ratio %.2f (=> %.0f MHz); dpp chain %.1f clk/op; exp chain %.1f clk/op; mfma dep chain %.1f clk/op; 2 indep mfma %.1f clk/pair; L2 chase %.0f clk/hop; total %.2f us\n",
           n, (double)h[0] / n, (double)h[0] / h[1], 100.0 * h[0] / h[1], (double)h[2] / n, (double)h[3] / n, (double)h[4] / n, (double)h[5] / n, (double)h[6] / 64, h[7] / 100.0);
  }
  return 0;
}
