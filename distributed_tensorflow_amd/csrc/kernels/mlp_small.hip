// Latency-optimised fp32 kernels for the reference 784-H-10 MLP
// (math: /root/reference/distributed.py:78-89 and its autodiff).
//
// The default workload is ~34 MFLOP and ~318 KB of parameters per step: it is
// bound by launch count and dependency latency, not by FLOPs.  The step is
// therefore three launches, each spread over many CUs with one 16x16 output
// tile per wave, operands loaded straight from L2 into MFMA fragment layout
// (no LDS round trip: nothing is re-used inside a workgroup), and the exact
// fp32 matrix instruction v_mfma_f32_16x16x4_f32:
//
//   1. small_gemm  : H = relu(X[rows] @ W1 + b1)          (split-K over the waves of a workgroup)
//   2. head        : z2, y, softmax(y), loss, accuracy, dz2, dz1   (one wave per 16 rows, all in registers)
//   3. wgrad       : dW1, db1, dW2, db2 as one grouped launch; optionally theta -= lr*grad in the epilogue,
//                    plus the step tail (counters, log record, next batch rows).
//
// Fragment maps of v_mfma_f32_16x16x4_f32 (lane l, fr = l & 15, g = l >> 4):
//   A operand: A[m = fr][k = g]     B operand: B[k = g][n = fr]
//   C/D      : reg r holds C[row = 4*g + r][col = fr]
#include "dtf_kernels.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Out-of-range operand elements are read from this zero block instead of being guarded:
// `ok ? *p : 0` is compiled into a branch around the load with its own s_waitcnt, which
// serialises a batch of loads that must all be in flight together (these kernels are
// latency bound).  Selecting the ADDRESS keeps every load unconditional.
__device__ float4 dtf_zero16[2];
__device__ __forceinline__ float ld_or_zero(const float* p, bool ok) {
  return *(ok ? p : reinterpret_cast<const float*>(dtf_zero16));
}
__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
  return *reinterpret_cast<const float4*>(ok ? p : reinterpret_cast<const float*>(dtf_zero16));
}
__device__ __forceinline__ int ldi_or_zero(const int* p, bool ok) {
  return *(ok ? p : reinterpret_cast<const int*>(dtf_zero16));
}

// --------------------------------------------------------------------------------------
// step tail
// --------------------------------------------------------------------------------------
// Runs in ONE workgroup of exactly 256 threads.
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

__global__ __launch_bounds__(256) void step_tail_kernel(DtfTailArgs t, int init_only) {
  dtf_tail_body(t, threadIdx.x, blockDim.x, init_only != 0);
}

DTF_EXPORT int dtf_step_tail(const DtfTailArgs* a, int init_only, hipStream_t s) {
  if (!a || !a->state) return DTF_ERR_ARG;
  if (a->ridx_next && (a->B <= 0 || a->n_train <= 0 || a->ridx_next == a->ridx_cur)) return DTF_ERR_SHAPE;
  if (!init_only && a->nparts > 0 && (!a->loss_part || !a->correct_part || !a->post_part)) return DTF_ERR_ARG;
  hipLaunchKernelGGL(step_tail_kernel, dim3(1), dim3(256), 0, s, *a, init_only);
  return dtf_check_launch();
}

// --------------------------------------------------------------------------------------
// small GEMM:  C = epilogue(A[rows] @ B)   with B either [K,N] or (NT) [N,K]
// --------------------------------------------------------------------------------------
template <int NW, bool BT, bool VEC>
__global__ __launch_bounds__(NW * 64) void small_gemm_kernel(DtfSmallGemmArgs p) {
  __shared__ float red[NW][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and provably so
  const int fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;

  const int arow = m0 + fr;
  const bool arow_ok = arow < p.M;
  const int asrc = p.a_rows ? ldi_or_zero(p.a_rows + arow, arow_ok) : (arow_ok ? arow : 0);
  const float* __restrict__ aptr = p.A + (size_t)asrc * p.lda;
  const int bcol = n0 + fr;
  const bool bcol_ok = bcol < p.N;
  const float* __restrict__ bptr = BT ? p.B + (size_t)(bcol_ok ? bcol : 0) * p.ldb : p.B + (bcol_ok ? bcol : 0);

  // Epilogue operands of the element this thread will finish (e = threadIdx.x, first 256 threads)
  // are requested before the main loop so their latency is not paid after the reduction.
  const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
  const int e_row = m0 + 4 * (e_ln >> 4) + e_r, e_col = n0 + (e_ln & 15);
  const bool e_ok = threadIdx.x < 256 && e_row < p.M && e_col < p.N;
  const float e_bias = p.bias ? ld_or_zero(p.bias + e_col, e_ok) : 0.f;
  const float e_mask = p.mask ? ld_or_zero(p.mask + (size_t)e_row * p.ldmask + e_col, e_ok) : 1.f;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nchunk = (p.K + 15) >> 4;
  // The kernel is latency bound: issue the operand loads of CH k-chunks back to back (fully
  // unrolled, guards are predicates not branches), then run their MFMAs.
  constexpr int CH = 8;
  for (int cb = wave; cb < nchunk; cb += NW * CH) {
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = ((cb + i * NW) << 4) + 4 * g;
      if (VEC) {
        const float4 t = ld4_or_zero(aptr + k, arow_ok && k < p.K);
        av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[i][j] = ld_or_zero(aptr + k + j, arow_ok && k + j < p.K);
      }
      if (BT) {
        if (VEC) {
          const float4 t = ld4_or_zero(bptr + k, bcol_ok && k < p.K);
          bv[i][0] = t.x; bv[i][1] = t.y; bv[i][2] = t.z; bv[i][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[i][j] = ld_or_zero(bptr + k + j, bcol_ok && k + j < p.K);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[i][j] = ld_or_zero(bptr + (size_t)(k + j) * p.ldb, bcol_ok && k + j < p.K);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      acc0 = MFMA16(av[i][0], bv[i][0], acc0);
      acc1 = MFMA16(av[i][1], bv[i][1], acc1);
      acc0 = MFMA16(av[i][2], bv[i][2], acc0);
      acc1 = MFMA16(av[i][3], bv[i][3], acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  __syncthreads();
  if (NW >= 4) {
    if (e_ok) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w][e_r][e_ln];
      s += e_bias;
      if (p.relu) s = fmaxf(s, 0.f);
      s = e_mask > 0.f ? s : 0.f;
      p.C[(size_t)e_row * p.ldc + e_col] = s;
    }
  } else {
    for (int e = threadIdx.x; e < 256; e += NW * 64) {
      const int r = e >> 6, ln = e & 63;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w][r][ln];
      const int row = m0 + 4 * (ln >> 4) + r;
      const int col = n0 + (ln & 15);
      if (row < p.M && col < p.N) {
        if (p.bias) s += p.bias[col];
        if (p.relu) s = fmaxf(s, 0.f);
        if (p.mask) s = p.mask[(size_t)row * p.ldmask + col] > 0.f ? s : 0.f;
        p.C[(size_t)row * p.ldc + col] = s;
      }
    }
  }
}

template <int NW>
static int launch_small_gemm(const DtfSmallGemmArgs& a, bool vec, hipStream_t s) {
  dim3 grid(dtf_cdiv(a.N, 16), dtf_cdiv(a.M, 16)), block(NW * 64);
  if (a.b_transposed) {
    if (vec) hipLaunchKernelGGL((small_gemm_kernel<NW, true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((small_gemm_kernel<NW, true, false>), grid, block, 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((small_gemm_kernel<NW, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((small_gemm_kernel<NW, false, false>), grid, block, 0, s, a);
  }
  return dtf_check_launch();
}

DTF_EXPORT int dtf_small_gemm(const DtfSmallGemmArgs* a, hipStream_t s) {
  if (!a || !a->A || !a->B || !a->C) return DTF_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return DTF_ERR_SHAPE;
  if (a->lda < a->K || a->ldc < a->N) return DTF_ERR_SHAPE;
  if (a->b_transposed ? a->ldb < a->K : a->ldb < a->N) return DTF_ERR_SHAPE;
  if (a->mask && a->ldmask < a->N) return DTF_ERR_SHAPE;
  if (dtf_cdiv(a->M, 16) > 65535) return DTF_ERR_SHAPE;
  // float4 operand loads need 16-byte aligned rows; otherwise take the scalar-load build.
  bool vec = (a->K % 4 == 0) && (a->lda % 4 == 0) && dtf_aligned16(a->A);
  if (a->b_transposed) vec = vec && (a->ldb % 4 == 0) && dtf_aligned16(a->B);
  // Deep reductions get 8 waves per tile so the dependent MFMA chain stays short.
  if (a->K >= 512) return launch_small_gemm<8>(*a, vec, s);
  if (a->K >= 64) return launch_small_gemm<4>(*a, vec, s);
  return launch_small_gemm<1>(*a, vec, s);
}

// --------------------------------------------------------------------------------------
// grouped weight gradient:  dW = X[rows]^T @ dY,  db = colsum(dY)   (+ fused SGD, + step tail)
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_grouped_kernel(DtfWgradArgs p) {
  const int nblk_tiles = (p.total_tiles + 3) >> 2;
  if ((int)blockIdx.x >= nblk_tiles) {  // one extra workgroup runs the step tail
    if (p.tail.enable) dtf_tail_body(p.tail, threadIdx.x, blockDim.x, false);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= p.total_tiles) return;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < DTF_MAX_WGRAD_PROBLEMS; ++i)
    if (i < p.nprob && tile >= p.prob[i].tile_start) pi = i;
  // pi is wave-uniform: the problem descriptor comes out of the kernarg segment by scalar loads
  const DtfWgradProblem q = p.prob[__builtin_amdgcn_readfirstlane(pi)];
  const int t = tile - q.tile_start;
  const int tn = t % q.tiles_n, tm = t / q.tiles_n;
  const int m0 = tm * 16, n0 = tn * 16;
  const int am = m0 + fr;
  const bool am_ok = am < q.Mo;
  const int bn = n0 + fr;
  const bool bn_ok = bn < q.N;
  const float* __restrict__ X = q.X;
  const float* __restrict__ dY = q.dY;
  const int* __restrict__ xrows = q.x_rows;

  // fused update: the old parameter values are requested up front, not after the MFMA chain
  float wold[4] = {0.f, 0.f, 0.f, 0.f};
  float bold = 0.f;
  if (p.mode == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      wold[r] = ld_or_zero(q.W + (size_t)row * q.ldw + bn, row < q.Mo && bn_ok);
    }
    bold = ld_or_zero(q.b + bn, bn_ok);
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  constexpr int CH = 8;   // 128 batch rows per pass: row indices, then every operand load, then the MFMAs
  for (int kb = 0; kb < q.Kb; kb += 16 * CH) {
    int xr[CH][4];
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + 16 * i + 4 * j + g;
        xr[i][j] = xrows ? ldi_or_zero(xrows + k, k < q.Kb) : k;
      }
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + 16 * i + 4 * j + g;
        const bool kok = k < q.Kb;
        av[i][j] = ld_or_zero(X + (size_t)xr[i][j] * q.ldx + am, kok && am_ok);
        bv[i][j] = ld_or_zero(dY + (size_t)k * q.ldy + bn, kok && bn_ok);
      }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      bsum += (bv[i][0] + bv[i][1]) + (bv[i][2] + bv[i][3]);
      acc0 = MFMA16(av[i][0], bv[i][0], acc0);
      acc1 = MFMA16(av[i][1], bv[i][1], acc1);
      acc0 = MFMA16(av[i][2], bv[i][2], acc0);
      acc1 = MFMA16(av[i][3], bv[i][3], acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * g + r;
    if (row < q.Mo && bn_ok) {
      const float gval = acc0[r] + acc1[r];
      const size_t off = (size_t)row * q.ldw + bn;
      if (p.mode == 1) q.W[off] = wold[r] - p.lr * gval;
      else q.dW[off] = gval;
    }
  }
  if (tm == 0) {  // wave-uniform: this tile also owns the bias gradient of its 16 columns
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (g == 0 && bn_ok) {
      if (p.mode == 1) q.b[bn] = bold - p.lr * bsum;
      else q.db[bn] = bsum;
    }
  }
}

DTF_EXPORT int dtf_wgrad_grouped(const DtfWgradArgs* a, hipStream_t s) {
  if (!a || a->nprob <= 0 || a->nprob > DTF_MAX_WGRAD_PROBLEMS) return DTF_ERR_ARG;
  int tiles = 0;
  for (int i = 0; i < a->nprob; ++i) {
    const DtfWgradProblem& q = a->prob[i];
    if (!q.X || !q.dY) return DTF_ERR_ARG;
    if (a->mode == 1 ? (!q.W || !q.b) : (!q.dW || !q.db)) return DTF_ERR_ARG;
    if (q.Mo <= 0 || q.N <= 0 || q.Kb <= 0) return DTF_ERR_SHAPE;
    if (q.ldx < q.Mo || q.ldy < q.N || q.ldw < q.N) return DTF_ERR_SHAPE;
    if (q.tiles_n != dtf_cdiv(q.N, 16) || q.tile_start != tiles) return DTF_ERR_SHAPE;
    tiles += q.tiles_n * dtf_cdiv(q.Mo, 16);
  }
  if (tiles != a->total_tiles) return DTF_ERR_SHAPE;
  if (a->tail.enable) {
    if (!a->tail.state) return DTF_ERR_ARG;
    if (a->tail.ridx_next && (a->tail.B <= 0 || a->tail.n_train <= 0 || a->tail.ridx_next == a->tail.ridx_cur)) return DTF_ERR_SHAPE;
    if (a->tail.nparts > 0 && (!a->tail.loss_part || !a->tail.correct_part || !a->tail.post_part)) return DTF_ERR_ARG;
  }
  const int nblk = dtf_cdiv(tiles, 4) + 1;
  hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(nblk), dim3(256), 0, s, *a);
  return dtf_check_launch();
}

// --------------------------------------------------------------------------------------
// head: logits, (double) softmax cross-entropy, accuracy, dz2 and dz1 for 16 rows per wave
// --------------------------------------------------------------------------------------
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1));
  v = fmaxf(v, __shfl_xor(v, 2));
  v = fmaxf(v, __shfl_xor(v, 4));
  v = fmaxf(v, __shfl_xor(v, 8));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}
// argmax with lowest-index tie break (tf.argmax semantics, distributed.py:83)
__device__ __forceinline__ int group16_argmax(float v, int idx) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    const float ov = __shfl_xor(v, m);
    const int oi = __shfl_xor(idx, m);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  return idx;
}

// One workgroup = 4 waves = 16 rows.  The three dependent phases are each split over the
// four waves so that the serial latency chain per wave stays short:
//   A  z2 partial sums, k-chunks dealt round-robin to the waves (reduced through LDS)
//   B  wave w does the softmax / loss / accuracy / dz2 math of rows 4g + w
//   C  dz1 column chunks dealt round-robin to the waves
__global__ __launch_bounds__(256) void head_kernel(DtfHeadArgs p) {
  __shared__ float red[4][4][64];
  __shared__ float dzs[16][17];
  __shared__ float part_loss[4];
  __shared__ int part_cnt[4][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 16;
  const int C = p.C, Hd = p.Hd;
  const float NEG_INF = -__builtin_huge_valf();
  const bool cls_ok = fr < C;
  const bool train_tile = r0 < p.M_train && p.dz1 != nullptr;   // workgroup-uniform
  const int nchunk = (Hd + 15) >> 4;
  constexpr int CH = 4;

  // ---- phase C operands of the first pass are requested first: they depend on nothing ----
  float cw[CH][4], cact[CH][4];
  auto load_c = [&](int cb) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int hcol = ((cb + 4 * i) << 4) + fr;
      const bool h_ok = hcol < Hd;
      const float* __restrict__ w2r = p.W2 + (size_t)hcol * C;
#pragma unroll
      for (int j = 0; j < 4; ++j) cw[i][j] = ld_or_zero(w2r + 4 * j + g, h_ok && 4 * j + g < C);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 4 * g + r;
        cact[i][r] = ld_or_zero(p.Hact + (size_t)row * p.ldh + hcol, h_ok && row < p.M_train);
      }
    }
  };
  if (train_tile) load_c(wave);

  // ---- phase A: z2 = Hact @ W2 (16 rows x 16 padded classes), split-K over the waves ----
  const int arow = r0 + fr;
  const bool arow_ok = arow < p.M_total;
  const float* __restrict__ hp = p.Hact + (size_t)arow * p.ldh;
  const float* __restrict__ wp = p.W2 + fr;
  const bool vec = ((p.ldh | Hd) & 3) == 0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int cb = wave; cb < nchunk; cb += 4 * CH) {
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = ((cb + 4 * i) << 4) + 4 * g;
      if (vec) {
        const float4 t = ld4_or_zero(hp + k, arow_ok && k < Hd);
        av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[i][j] = ld_or_zero(hp + k + j, arow_ok && k + j < Hd);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[i][j] = ld_or_zero(wp + (size_t)(k + j) * C, cls_ok && k + j < Hd);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      acc0 = MFMA16(av[i][0], bv[i][0], acc0);
      acc1 = MFMA16(av[i][1], bv[i][1], acc1);
      acc0 = MFMA16(av[i][2], bv[i][2], acc0);
      acc1 = MFMA16(av[i][3], bv[i][3], acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  // label / bias requests overlap the barrier
  const int row = r0 + 4 * g + wave;            // the one row this lane group handles in phase B
  const bool row_ok = row < p.M_total;
  const bool train_row = row < p.M_train;
  const int lrow = p.rows ? ldi_or_zero(p.rows + row, row_ok) : row;
  const float lab = ld_or_zero(p.labels + (size_t)lrow * C + fr, row_ok && cls_ok);
  const float bias = ld_or_zero(p.b2 + fr, cls_ok);
  __syncthreads();

  // ---- phase B ----
  float loss_acc = 0.f;
  int correct_acc = 0, post_acc = 0;
  {
    const float zsum = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
    const float z = cls_ok ? zsum + bias : NEG_INF;
    const float zmax = group16_max(z);
    const float e1 = cls_ok ? expf(z - zmax) : 0.f;
    const float s1 = group16_sum(e1);
    const float y = e1 / s1;
    if (p.probs && row_ok && cls_ok) p.probs[(size_t)row * C + fr] = y;
    float logp, dz;
    if (p.loss_mode == 0) {
      // softmax applied a second time on the probabilities (the reference feeds y as "logits")
      const float ymax = group16_max(cls_ok ? y : NEG_INF);
      const float e2 = cls_ok ? expf(y - ymax) : 0.f;
      const float s2 = group16_sum(e2);
      const float pp = e2 / s2;
      logp = (y - ymax) - logf(s2);
      const float dy = (pp - lab) * p.inv_b;
      const float dot = group16_sum(cls_ok ? dy * y : 0.f);
      dz = (dy - dot) * y;
    } else {
      logp = (z - zmax) - logf(s1);
      dz = (y - lab) * p.inv_b;
    }
    const float lrow_loss = -group16_sum(cls_ok ? lab * logp : 0.f);
    const int am_y = group16_argmax(cls_ok ? y : NEG_INF, fr);
    const int am_l = group16_argmax(cls_ok ? lab : NEG_INF, fr);
    const int hit = (row_ok && am_y == am_l) ? 1 : 0;
    if (fr == 0) {
      if (train_row) { loss_acc = lrow_loss; correct_acc = hit; }
      else post_acc = hit;
    }
    const float dzv = (train_row && cls_ok) ? dz : 0.f;
    dzs[4 * g + wave][fr] = dzv;
    if (train_row && p.dz2) p.dz2[(size_t)row * 16 + fr] = dzv;
  }
  loss_acc += __shfl_xor(loss_acc, 16);
  loss_acc += __shfl_xor(loss_acc, 32);
  correct_acc += __shfl_xor(correct_acc, 16);
  correct_acc += __shfl_xor(correct_acc, 32);
  post_acc += __shfl_xor(post_acc, 16);
  post_acc += __shfl_xor(post_acc, 32);
  if (lane == 0) {
    part_loss[wave] = loss_acc;
    part_cnt[wave][0] = correct_acc;
    part_cnt[wave][1] = post_acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // rows are summed in a fixed order: the logged loss is reproducible
    if (p.loss_part) p.loss_part[blockIdx.x] = (part_loss[0] + part_loss[1]) + (part_loss[2] + part_loss[3]);
    if (p.correct_part) p.correct_part[blockIdx.x] = part_cnt[0][0] + part_cnt[1][0] + part_cnt[2][0] + part_cnt[3][0];
    if (p.post_part) p.post_part[blockIdx.x] = part_cnt[0][1] + part_cnt[1][1] + part_cnt[2][1] + part_cnt[3][1];
  }
  if (!train_tile) return;

  // ---- phase C: dz1 = (dz2 @ W2^T) * (Hact > 0), 16 columns per chunk ----
  float a4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = dzs[fr][4 * j + g];
  for (int cb = wave; cb < nchunk; cb += 4 * CH) {
    if (cb != wave) load_c(cb);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int hcol = ((cb + 4 * i) << 4) + fr;
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = MFMA16(a4[0], cw[i][0], d0);
      d1 = MFMA16(a4[1], cw[i][1], d1);
      d0 = MFMA16(a4[2], cw[i][2], d0);
      d1 = MFMA16(a4[3], cw[i][3], d1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = r0 + 4 * g + r;
        if (orow < p.M_train && hcol < Hd)
          p.dz1[(size_t)orow * p.ld_dz1 + hcol] = cact[i][r] > 0.f ? d0[r] + d1[r] : 0.f;
      }
    }
  }
}

DTF_EXPORT int dtf_head(const DtfHeadArgs* a, hipStream_t s) {
  if (!a || !a->Hact || !a->W2 || !a->b2 || !a->labels) return DTF_ERR_ARG;
  if (a->M_total <= 0 || a->M_train < 0 || a->M_train > a->M_total) return DTF_ERR_SHAPE;
  if (a->C <= 0 || a->C > 16 || a->Hd <= 0 || a->ldh < a->Hd) return DTF_ERR_SHAPE;
  if (a->M_train > 0 && a->dz1 && a->ld_dz1 < a->Hd) return DTF_ERR_SHAPE;
  if (((a->ldh | a->Hd) & 3) == 0 && !dtf_aligned16(a->Hact)) return DTF_ERR_ALIGN;
  hipLaunchKernelGGL(head_kernel, dim3(dtf_cdiv(a->M_total, 16)), dim3(256), 0, s, *a);
  return dtf_check_launch();
}
