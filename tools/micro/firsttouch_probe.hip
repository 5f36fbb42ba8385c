// Micro-probe: latency of the first vector loads of a kernel (data produced by the previous launch), same
// page vs a new page, and of a scalar load, in shader clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void producer(float* a, float* b, float* c) {
  a[threadIdx.x] = 1.f; a[65536 + threadIdx.x] = 2.f; a[1024 + threadIdx.x] = 5.f;
  b[threadIdx.x] = 3.f; b[65536 + threadIdx.x] = 4.f; c[threadIdx.x] = 6.f;
}
__global__ void probe(unsigned long long* out, const float* a, const float* b, const float* c, const int* sc) {
  unsigned long long t[8];
  float acc = 0.f;
  t[0] = __builtin_amdgcn_s_memtime();
  acc += a[threadIdx.x];                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t[1] = __builtin_amdgcn_s_memtime();
  acc += a[1024 + threadIdx.x];          asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t[2] = __builtin_amdgcn_s_memtime();
  acc += a[65536 + threadIdx.x];         asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t[3] = __builtin_amdgcn_s_memtime();
  acc += b[threadIdx.x];                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t[4] = __builtin_amdgcn_s_memtime();
  acc += b[65536 + threadIdx.x];         asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t[5] = __builtin_amdgcn_s_memtime();
  acc += c[threadIdx.x];                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t[6] = __builtin_amdgcn_s_memtime();
  int s = sc[blockIdx.x];                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t[7] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) for (int i = 0; i < 7; ++i) out[i] = t[i + 1] - t[i];
  if (acc == 123.f) out[8] = s;
}
int main() {
  unsigned long long* out; float *a, *b, *c; int* sc;
  (void)hipMalloc(&out, 128); (void)hipMalloc(&a, 64 << 20); (void)hipMalloc(&b, 64 << 20); (void)hipMalloc(&c, 4096); (void)hipMalloc(&sc, 4096);
  (void)hipMemset(sc, 0, 4096);
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, 0, a, b, c);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, a, b, c, sc);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("first load %llu clk | +4KB same alloc %llu | +256KB same alloc %llu | other alloc %llu | +256KB in it %llu | small alloc %llu | scalar load %llu\n",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
  }
  return 0;
}
