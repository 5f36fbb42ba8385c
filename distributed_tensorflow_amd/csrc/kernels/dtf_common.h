// Shared declarations for the gfx950 (MI355X / CDNA4) kernel library.
//
// Everything in this library is written for one target: 64-lane wavefronts,
// MFMA matrix cores, 160 KiB LDS per CU, 8 XCDs x 32 CUs.  There is no second
// code path.  The exported C ABI takes raw device pointers plus a hipStream_t
// so that the Python layer (ctypes) can launch on whatever stream PyTorch is
// currently using, including a stream that is being captured into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DTF_EXPORT extern "C" __attribute__((visibility("default")))

// Error codes returned by every launcher (0 = launched).
enum {
  DTF_OK = 0,
  DTF_ERR_SHAPE = 1,     // operand shapes do not match what the kernel's grid assumes
  DTF_ERR_ALIGN = 2,     // pointer or leading dimension violates the 16-byte rule
  DTF_ERR_LAUNCH = 3,    // hipGetLastError() reported a failure
  DTF_ERR_ARG = 4,       // null pointer / unsupported option
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

static inline int dtf_cdiv(int a, int b) { return (a + b - 1) / b; }

static inline int dtf_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DTF_OK : DTF_ERR_LAUNCH;
}

static inline bool dtf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Bijective XCD-aware remap of a linear workgroup id.  Workgroups are dealt
// round-robin over the 8 XCDs (each with a private L2), so consecutive ids land
// on different L2s.  After the remap, ids that share an XCD cover a contiguous
// range of tiles and re-use each other's operand panels from that XCD's L2.
__device__ __forceinline__ int dtf_xcd_remap(int orig, int nwg) {
  const int NX = 8;
  if (nwg < NX * 2) return orig;
  int q = nwg / NX, r = nwg % NX;
  int xcd = orig % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / NX;
}


// One Adam element update, the single statement of it that every kernel uses (flat optimizer, ranges, the folded
// weight-gradient epilogue, the async push): lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) and eps_t = eps * sqrt(1 - b2^t)
// are computed once per launch.  Square root and reciprocal are the hardware instructions (v_sqrt_f32, v_rcp_f32, 1 ulp
// each): the IEEE-exact division and square root cost ~25 instructions per element, which made the folded epilogue
// (128 elements per lane) spend as long in the vector ALU as in memory; 2 ulp on a step of relative size lr is far
// below the rounding of the parameter itself.
__device__ __forceinline__ void dtf_adam_one(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps_t) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  p -= lr_t * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps_t);
}
