// Micro-probe: cost of cold instruction fetch.  The same 4096 VALU operations run (a) as one straight-line
// stream (32 KB of code, every line fetched once, cold after the kernel boundary) and (b) as a 64-instruction
// loop body executed 64 times (fetched once, then served by the instruction cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R1024(x) R16(R64(x))
__global__ void straight(unsigned long long* out, float* data) {
  float a = data[threadIdx.x], b = a + 1.f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  R1024(asm volatile("v_fmac_f32 %0, %1, %1\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %1, %0, %0" : "+v"(a), "+v"(b));)
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
  data[64 + threadIdx.x] = a + b;
}
__global__ void looped(unsigned long long* out, float* data) {
  float a = data[threadIdx.x], b = a + 1.f;
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) {
    R16(asm volatile("v_fmac_f32 %0, %1, %1\n v_fmac_f32 %1, %0, %0\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %1, %0, %0" : "+v"(a), "+v"(b));)
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
  data[64 + threadIdx.x] = a + b;
}
int main() {
  unsigned long long* out; float* data;
  (void)hipMalloc(&out, 8 * 256); (void)hipMalloc(&data, 4096); (void)hipMemset(data, 0, 4096);
  unsigned long long h[256];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(straight, dim3(1), dim3(64), 0, 0, out, data); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    unsigned long long s1 = h[0];
    hipLaunchKernelGGL(looped, dim3(1), dim3(64), 0, 0, out, data); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("4096 dependent VALU ops: straight-line (16 KB of code) %llu clk = %.2f us; 64x loop of 64 %llu clk = %.2f us\n", s1, s1 / 2400.0, h[0], h[0] / 2400.0);
  }
  return 0;
}
