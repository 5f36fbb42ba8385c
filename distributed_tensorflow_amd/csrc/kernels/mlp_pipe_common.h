// Pieces shared by the one-launch pipelined step of one worker (mlp_pipe.hip) and of N synchronous workers
// (mlp_pipe_multi.hip): slot / role constants, the fragment-order storage maps, the 16-lane softmax head.
#pragma once
#include "dtf_device.h"

// raw buffer load with a wave-uniform (scalar) offset added to the per-lane one; the range check covers the per-lane
// offset only, so kOOB there returns 0 whatever the scalar part is
__device__ __forceinline__ float pipe_bld_s(rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ void pipe_bld4_s(rsrc_t r, unsigned voff, unsigned soff, float (&v)[4]) {
  auto t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
}

#define PIPE_SLOTS 8
#define PIPE_ROWS 128
#define PIPE_LABG_SLOT (16 * PIPE_ROWS)
#define PIPE_ROLE_L 1
#define PIPE_ROLE_Q 2
#define PIPE_ROLE_G 4
#define PIPE_ROLE_W 8
#define PIPE_ROLE_T 16
#define PIPE_ROLE_F 32
#define PIPE_ROLE_C 64              // prime launch: working copies from the flat parameters, pending gradients zero

// Operand-layout storage: D1, the Gram matrices, Q and Z1 are only ever read as MFMA operands, so they are stored the
// way a wave reads them - per 16x16 tile (8 x 8 tiles per matrix) 64 lanes x 4 consecutive values - and a fragment
// load is one 16-byte load per lane, 1 KB contiguous per wave (row-major storage made every such load touch 4-16
// separate cache lines, which cost the row workgroups 1.3 us per launch).
//   B operand (D1: k = batch row, n = hidden column):   tile (n / 16, k / 16), lane = ((k % 16) / 4) * 16 + n % 16, e = k % 4
//   A operand (Gram: i = row, k = column):              tile (i / 16, k / 16), lane = ((k % 16) / 4) * 16 + i % 16, e = k % 4
//   C layout  (Q, Z1: MFMA result, row i, column n):    tile (i / 16, n / 16), lane = ((i % 16) / 4) * 16 + n % 16, e = i % 4
//   output layer (W2m, P2: row k = hidden index, column c = class): per 16 rows 64 lanes x 4 values,
//                                                        lane = ((k % 16) / 4) * 16 + c, e = k % 4
//     = the MFMA result layout of h'^T dz2 (what a row workgroup produces) and, with MFMA step j taking k = 4g + j,
//       the B-operand layout of the logits product (what the next launch's row workgroups consume): 16-byte loads
//   working copy of W1' (Wm: k = input index, n = hidden column): B operand of the Q products and MFMA result of the
//     weight update alike, tile (k / 16, n / 16) of KT x NT tiles, lane = ((k % 16) / 4) * 16 + n % 16, e = k % 4
#define PIPE_MAT 16384                    // floats per operand-layout matrix (8 x 8 tiles x 256)
__host__ __device__ __forceinline__ int pipe_w2_off(int k, int c) { return ((k >> 4) * 64 + ((k & 15) >> 2) * 16 + c) * 4 + (k & 3); }
__host__ __device__ __forceinline__ unsigned pipe_frag(int t0, int t1, int lane) { return (unsigned)(((t0 * 8 + t1) * 64 + lane) * 4); }
__host__ __device__ __forceinline__ unsigned pipe_wfrag(int kt, int nt, int NT, int lane) { return (unsigned)(((kt * NT + nt) * 64 + lane) * 4); }

// sum of the 16 waves' partial accumulators: thread t < 256 owns one element of the 16x16 tile
__device__ __forceinline__ float pipe_reduce16(float (*red)[4][64], int e_r, int e_ln) {
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += red[w][e_r][e_ln];
  return s;
}

// softmax / loss / gradient of one row spread over 16 lanes (lane & 15 = class), the formulas of the two-launch
// step (distributed.py:81-87 and the autodiff of :102); the second softmax works on probabilities (values in
// [0, 1]) and needs no maximum; `lab_class` = argmax of the label row, found when the row was gathered
__device__ __forceinline__ void pipe_row_head(float zsum, float lab, int lab_class, bool cls_ok, bool grad_row, int fr,
                                              int loss_mode, float inv_b, float& loss, int& hit, float& dz) {
  const float NEG_INF = -__builtin_huge_valf();
  const float z = cls_ok ? zsum : NEG_INF;
  const float zmax = row16_max(z);
  const float e1 = __expf(z - zmax);                 // 0 on the padding lanes
  const float s1 = row16_sum(e1);
  const float inv1 = __builtin_amdgcn_rcpf(s1);
  float logp, gz;
  if (loss_mode == 0) {
    // softmax applied a second time on the probabilities (the reference feeds y as "logits")
    const float y = e1 * inv1;
    const float e2 = cls_ok ? __expf(y) : 0.f;
    const float s2 = row16_sum(e2);
    const float inv2 = __builtin_amdgcn_rcpf(s2);
    logp = y - __logf(s2);
    const float dy = cls_ok ? (e2 * inv2 - lab) * inv_b : 0.f;
    const float dot = row16_sum(dy * y);
    gz = (dy - dot) * y;
  } else {
    logp = (z - zmax) - __logf(s1);
    gz = (e1 * inv1 - lab) * inv_b;
  }
  loss = -row16_sum(cls_ok ? lab * logp : 0.f);
  const int am_y = row16_min((cls_ok && z == zmax) ? fr : 16);
  hit = am_y == lab_class ? 1 : 0;
  dz = (grad_row && cls_ok) ? gz : 0.f;
}

