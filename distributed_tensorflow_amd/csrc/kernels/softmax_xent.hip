// Fused softmax cross-entropy for any number of classes: loss, gradient wrt the logits and the argmax
// accuracy count in one pass over the logits, fp32 throughout.
//
// Math (the same two loss modes as the small-model head kernel):
//   standard   p = softmax(z); L = -sum_c lab_c log p_c;              dz = (p - lab) / B
//   reference  y = softmax(z); p = softmax(y) (the script feeds the softmax output as "logits",
//              /root/reference/distributed.py:81 + :86); L = -sum_c lab_c log p_c;
//              dy = (p - lab) / B;  dz = (dy - sum_c dy_c y_c) * y      (SoftmaxGrad through the first softmax)
//   accuracy   argmax_c z == argmax_c lab, lowest index on ties (tf.argmax, distributed.py:83-84)
// Labels are one-hot / soft rows [B, C] (float) or class indices [B] (int64).
//
// One batch row per 64-lane wave, four rows per workgroup.  Lane l owns classes l, l+64, ...: a row is read
// ONCE with coalesced 256-byte wave loads and kept in registers (up to 32 values per lane = 2048 classes;
// wider rows are re-read from L2 in the later passes).  Row-wide maxima and sums are wave reductions by DPP
// row rotations plus two cross-row swizzles; no LDS, no barrier, no atomics.  Per-workgroup partial loss
// sums and hit counts are folded by the caller in a fixed order (reproducible loss).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include "dtf_device.h"

namespace {
constexpr int kRowsPerBlock = 4;
constexpr int kMaxRegElems = 32;               // classes per lane kept in registers, at most

__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);                            // every lane of a 16-lane row holds the row's sum
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
  v = row16_min(v);
  v = min(v, __shfl_xor(v, 16));
  v = min(v, __shfl_xor(v, 32));
  return v;
}

// `lrow`: row of the label arrays that belongs to batch row `row` (an index gather when label_rows is given)
__device__ __forceinline__ float lab_at(const DtfXentArgs& a, int lrow, int c, long long cls) {
  return a.labels_f ? a.labels_f[(size_t)lrow * a.ld_labels + c] : (c == cls ? 1.f : 0.f);
}

// NE: classes per lane held in registers (C <= 64 * NE); 0 = stream the row from memory in every pass.
template <int NE>
__global__ __launch_bounds__(64 * kRowsPerBlock) void softmax_xent_kernel(DtfXentArgs a) {
  constexpr bool IN_REGS = NE > 0;
  constexpr int kRegElems = NE > 0 ? NE : 1;
  __shared__ float s_loss[kRowsPerBlock];
  __shared__ int s_hit[kRowsPerBlock];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = blockIdx.x * kRowsPerBlock + wave;
  const bool row_ok = row < a.B;
  const int C = a.C;
  const float NEG_INF = -__builtin_huge_valf();
  float loss = 0.f;
  int hit = 0;
  if (row_ok) {
    const float* z = a.logits + (size_t)row * a.ld_logits;
    const int lrow = a.label_rows ? a.label_rows[row] : row;
    const long long cls = a.labels_i ? a.labels_i[lrow] : -1;
    float zr[kRegElems];
    // ---- pass 1: row maximum and its first position ----
    float zmax = NEG_INF;
    if (IN_REGS) {
#pragma unroll
      for (int j = 0; j < kRegElems; ++j) {
        const int c = lane + 64 * j;
        zr[j] = c < C ? z[c] : NEG_INF;
        zmax = fmaxf(zmax, zr[j]);
      }
    } else {
      for (int c = lane; c < C; c += 64) zmax = fmaxf(zmax, z[c]);
    }
    zmax = wave_max(zmax);
    // ---- pass 2: sum of exponentials; argmax of logits and of labels ----
    float s1 = 0.f, labmax = NEG_INF;
    int am_z = 0x7fffffff;
    if (IN_REGS) {
#pragma unroll
      for (int j = 0; j < kRegElems; ++j) {
        const int c = lane + 64 * j;
        if (c < C) {
          s1 += __expf(zr[j] - zmax);
          if (zr[j] == zmax) am_z = min(am_z, c);
          if (a.labels_f) labmax = fmaxf(labmax, a.labels_f[(size_t)lrow * a.ld_labels + c]);
        }
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float v = z[c];
        s1 += __expf(v - zmax);
        if (v == zmax) am_z = min(am_z, c);
        if (a.labels_f) labmax = fmaxf(labmax, a.labels_f[(size_t)lrow * a.ld_labels + c]);
      }
    }
    s1 = wave_sum(s1);
    am_z = wave_min(am_z);
    int am_l;
    if (a.labels_f) {
      labmax = wave_max(labmax);
      int cand = 0x7fffffff;
      for (int c = lane; c < C; c += 64)
        if (a.labels_f[(size_t)lrow * a.ld_labels + c] == labmax) { cand = c; break; }
      am_l = wave_min(cand);
    } else {
      am_l = (int)cls;
    }
    hit = am_z == am_l ? 1 : 0;
    const float inv1 = 1.f / s1, log_s1 = __logf(s1);
    float* dz = a.dlogits ? a.dlogits + (size_t)row * a.ld_dlogits : nullptr;
    __hip_bfloat16* dzb = a.dlogits_bf ? reinterpret_cast<__hip_bfloat16*>(a.dlogits_bf) + (size_t)row * a.ld_dlogits_bf : nullptr;
    if (a.loss_mode == 1) {
      // ---- standard cross-entropy on the logits ----
      float l = 0.f;
      auto body = [&](int c, float v) {
        const float lab = lab_at(a, lrow, c, cls);
        l -= lab * ((v - zmax) - log_s1);
        const float g = (__expf(v - zmax) * inv1 - lab) * a.grad_scale;
        if (dz) dz[c] = g;
        if (dzb) dzb[c] = __float2bfloat16(g);
      };
      if (IN_REGS) {
#pragma unroll
        for (int j = 0; j < kRegElems; ++j) {
          const int c = lane + 64 * j;
          if (c < C) body(c, zr[j]);
        }
      } else {
        for (int c = lane; c < C; c += 64) body(c, z[c]);
      }
      loss = wave_sum(l);
    } else {
      // ---- the script's loss: softmax applied a second time on the probabilities ----
      // y in (0, 1]: its maximum is exp(0) / s1, no second max pass needed
      const float ymax = inv1;
      float s2 = 0.f;
      if (IN_REGS) {
#pragma unroll
        for (int j = 0; j < kRegElems; ++j) {
          const int c = lane + 64 * j;
          if (c < C) s2 += __expf(__expf(zr[j] - zmax) * inv1 - ymax);
        }
      } else {
        for (int c = lane; c < C; c += 64) s2 += __expf(__expf(z[c] - zmax) * inv1 - ymax);
      }
      s2 = wave_sum(s2);
      const float inv2 = 1.f / s2, log_s2 = __logf(s2);
      float l = 0.f, dot = 0.f;
      auto first = [&](int c, float v) {
        const float y = __expf(v - zmax) * inv1;
        const float lab = lab_at(a, lrow, c, cls);
        l -= lab * ((y - ymax) - log_s2);
        dot += (__expf(y - ymax) * inv2 - lab) * a.grad_scale * y;
      };
      if (IN_REGS) {
#pragma unroll
        for (int j = 0; j < kRegElems; ++j) {
          const int c = lane + 64 * j;
          if (c < C) first(c, zr[j]);
        }
      } else {
        for (int c = lane; c < C; c += 64) first(c, z[c]);
      }
      loss = wave_sum(l);
      dot = wave_sum(dot);
      if (dz || dzb) {
        auto second = [&](int c, float v) {
          const float y = __expf(v - zmax) * inv1;
          const float dy = (__expf(y - ymax) * inv2 - lab_at(a, lrow, c, cls)) * a.grad_scale;
          const float g = (dy - dot) * y;
          if (dz) dz[c] = g;
          if (dzb) dzb[c] = __float2bfloat16(g);
        };
        if (IN_REGS) {
#pragma unroll
          for (int j = 0; j < kRegElems; ++j) {
            const int c = lane + 64 * j;
            if (c < C) second(c, zr[j]);
          }
        } else {
          for (int c = lane; c < C; c += 64) second(c, z[c]);
        }
      }
    }
    if (a.loss_rows && lane == 0) a.loss_rows[row] = loss;
  }
  if (lane == 0) {
    s_loss[wave] = row_ok ? loss : 0.f;
    s_hit[wave] = row_ok ? hit : 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (a.loss_part) a.loss_part[blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
    if (a.correct_part) a.correct_part[blockIdx.x] = s_hit[0] + s_hit[1] + s_hit[2] + s_hit[3];
  }
}
}  // namespace

DTF_EXPORT int dtf_softmax_xent_blocks(int B) { return dtf_cdiv(B, kRowsPerBlock); }

DTF_EXPORT int dtf_softmax_xent(const DtfXentArgs* a, hipStream_t s) {
  if (!a || !a->logits || (!a->labels_f && !a->labels_i) || (a->labels_f && a->labels_i)) return DTF_ERR_ARG;
  if (a->B <= 0 || a->C <= 0 || a->ld_logits < a->C) return DTF_ERR_SHAPE;
  if (a->labels_f && a->ld_labels < a->C) return DTF_ERR_SHAPE;
  if (a->dlogits && a->ld_dlogits < a->C) return DTF_ERR_SHAPE;
  if (a->dlogits_bf && a->ld_dlogits_bf < a->C) return DTF_ERR_SHAPE;
  if (a->loss_mode != 0 && a->loss_mode != 1) return DTF_ERR_ARG;
  const int grid = dtf_cdiv(a->B, kRowsPerBlock);
  const dim3 g(grid), b(64 * kRowsPerBlock);
  if (a->C <= 64) hipLaunchKernelGGL(softmax_xent_kernel<1>, g, b, 0, s, *a);
  else if (a->C <= 256) hipLaunchKernelGGL(softmax_xent_kernel<4>, g, b, 0, s, *a);
  else if (a->C <= 1024) hipLaunchKernelGGL(softmax_xent_kernel<16>, g, b, 0, s, *a);
  else if (a->C <= 64 * kMaxRegElems) hipLaunchKernelGGL(softmax_xent_kernel<kMaxRegElems>, g, b, 0, s, *a);
  else hipLaunchKernelGGL(softmax_xent_kernel<0>, g, b, 0, s, *a);
  return dtf_check_launch();
}
