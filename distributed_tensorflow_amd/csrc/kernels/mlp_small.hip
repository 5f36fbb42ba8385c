// Latency-optimised fp32 kernels for the reference 784-H-10 MLP
// (math: /root/reference/distributed.py:78-89 and its autodiff).
//
// The default workload is ~34 MFLOP and ~318 KB of parameters per step: it is
// bound by launch count and dependency latency, not by FLOPs.  The step is
// three launches, each spread over many CUs with one 16x16 output tile per
// wave, operands loaded straight from L2 into MFMA fragment layout (nothing is
// re-used inside a workgroup, so no LDS staging), and the exact fp32 matrix
// instruction v_mfma_f32_16x16x4_f32:
//
//   1. small_gemm  : H = relu(X[rows] @ W1 + b1)          (split-K over the waves of a workgroup)
//   2. head        : z2, y, softmax(y), loss, accuracy, dz2, dz1   (16 rows per workgroup, in registers)
//   3. wgrad       : dW1, db1, dW2, db2 as one grouped launch; optionally theta -= lr*grad in the epilogue,
//                    plus the step tail (counters, log record, next batch rows).
//
// What was measured while tuning these (hipGraph replay, MI355X):
//   * An empty kernel costs 1.9 us per graph node.  A kernel's instruction stream is fetched cold
//     after every kernel boundary, so its straight-line length is paid in latency: the same GEMM
//     ran 4.2 us with an 8-chunk unrolled body and 2.6 us with a 1-chunk body at K = 64.
//     Hence: short bodies, loops that re-execute cached code, few and wide memory instructions.
//   * `ok ? *p : 0` becomes a branch with its own s_waitcnt and serialises a batch of loads.
//     All operand loads are therefore raw buffer loads: the hardware range check returns 0 for an
//     out-of-range offset, so guards are a select on a 32-bit offset, never a branch.
//   * Reductions over the 16 classes use DPP row rotations (one VALU op per step) instead of
//     ds_bpermute shuffles.
//
// Fragment maps of v_mfma_f32_16x16x4_f32 (lane l, fr = l & 15, g = l >> 4):
//   A operand: A[m = fr][k = g]     B operand: B[k = g][n = fr]
//   C/D      : reg r holds C[row = 4*g + r][col = fr]
#include "dtf_device.h"

// Diagnostic build only (-DDTF_STAMPS): per-workgroup timestamps of the 100 MHz constant clock,
// written to a buffer of their own.  No stamp executes in the production library.
#ifdef DTF_STAMPS
__device__ unsigned long long* dtf_stamp_buf = nullptr;
#define DTF_STAMP(slot, wg)                                                                                             \
  do {                                                                                                                  \
    if (threadIdx.x == 0 && dtf_stamp_buf) dtf_stamp_buf[(size_t)(wg) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int dtf_set_stamp_buffer(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dtf_stamp_buf), &p, sizeof(p));
}
__global__ void stamp_now_kernel(unsigned long long* out) { *out = __builtin_amdgcn_s_memrealtime(); }
extern "C" __attribute__((visibility("default"))) int dtf_stamp_now(unsigned long long* out, hipStream_t s) {
  hipLaunchKernelGGL(stamp_now_kernel, dim3(1), dim3(1), 0, s, out);
  return 0;
}
#else
#define DTF_STAMP(slot, wg) do { } while (0)
#endif

// --------------------------------------------------------------------------------------
// step tail (body in dtf_device.h).  Runs in ONE workgroup of exactly 256 threads.
// --------------------------------------------------------------------------------------
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
// One 16x16 tile per workgroup; the NW waves split K round-robin in 16-deep chunks.
// --------------------------------------------------------------------------------------
#ifndef DTF_GEMM_CH
#define DTF_GEMM_CH 4
#endif

template <int NW, bool BT, bool VEC>
__global__ __launch_bounds__(NW * 64) void small_gemm_kernel(DtfSmallGemmArgs p) {
  __shared__ float red[NW][4][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and provably so
  const int fr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int wg_id = blockIdx.y * gridDim.x + blockIdx.x;
  (void)wg_id;
  DTF_STAMP(0, wg_id);

  const rsrc_t ra = make_rsrc(p.A, p.a_bytes), rb = make_rsrc(p.B, p.b_bytes);
  const int arow = m0 + fr;
  const bool arow_ok = arow < p.M;
  const int asrc = p.a_rows ? p.a_rows[arow_ok ? arow : 0] : arow;
  const unsigned a_row_off = arow_ok ? (unsigned)asrc * (unsigned)p.lda * 4u : kOOB;
  const int bcol = n0 + fr;
  const bool bcol_ok = bcol < p.N;
  const unsigned b_lane_off = bcol_ok ? (BT ? (unsigned)bcol * (unsigned)p.ldb * 4u : (unsigned)bcol * 4u) : kOOB;

  // Epilogue operands of the element this thread will finish (first 256 threads) are requested
  // before the main loop so their latency is not paid after the reduction.
  const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
  const int e_row = m0 + 4 * (e_ln >> 4) + e_r, e_col = n0 + (e_ln & 15);
  const bool e_ok = threadIdx.x < 256 && e_row < p.M && e_col < p.N;
  const float e_bias = (p.bias && e_ok) ? p.bias[e_col] : 0.f;
  const float e_mask = (p.mask && e_ok) ? p.mask[(size_t)e_row * p.ldmask + e_col] : 1.f;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nchunk = (p.K + 15) >> 4;
  constexpr int CH = DTF_GEMM_CH;
  for (int cb = wave; cb < nchunk; cb += NW * CH) {
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = ((cb + i * NW) << 4) + 4 * g;   // chunks past the end: every k >= K, every offset out of range
      if (VEC) {
        bld4(ra, (a_row_off != kOOB && k < p.K) ? a_row_off + (unsigned)k * 4u : kOOB, av[i]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[i][j] = bld(ra, (a_row_off != kOOB && k + j < p.K) ? a_row_off + (unsigned)(k + j) * 4u : kOOB);
      }
      if (BT) {
        if (VEC) {
          bld4(rb, (b_lane_off != kOOB && k < p.K) ? b_lane_off + (unsigned)k * 4u : kOOB, bv[i]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[i][j] = bld(rb, (b_lane_off != kOOB && k + j < p.K) ? b_lane_off + (unsigned)(k + j) * 4u : kOOB);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bv[i][j] = bld(rb, (b_lane_off != kOOB && k + j < p.K) ? b_lane_off + (unsigned)(k + j) * (unsigned)p.ldb * 4u : kOOB);
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
  DTF_STAMP(1, wg_id);
  if (NW == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      if (row < p.M && bcol_ok) {
        float s = acc0[r] + acc1[r];
        if (p.bias) s += p.bias[bcol];
        if (p.relu) s = fmaxf(s, 0.f);
        if (p.mask) s = p.mask[(size_t)row * p.ldmask + bcol] > 0.f ? s : 0.f;
        p.C[(size_t)row * p.ldc + bcol] = s;
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  __syncthreads();
  DTF_STAMP(2, wg_id);
  if (e_ok) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w][e_r][e_ln];
    s += e_bias;
    if (p.relu) s = fmaxf(s, 0.f);
    s = e_mask > 0.f ? s : 0.f;
    p.C[(size_t)e_row * p.ldc + e_col] = s;
  }
  DTF_STAMP(3, wg_id);
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
  // operands are addressed with 32-bit byte offsets through buffer descriptors
  if (a->a_bytes == 0 || a->b_bytes == 0 || a->a_bytes >= kOOB || a->b_bytes >= kOOB) return DTF_ERR_SHAPE;
  if (!a->a_rows && (uint64_t)((uint64_t)(a->M - 1) * a->lda + a->K) * 4u > a->a_bytes) return DTF_ERR_SHAPE;
  const uint64_t b_need = a->b_transposed ? ((uint64_t)(a->N - 1) * a->ldb + a->K) : ((uint64_t)(a->K - 1) * a->ldb + a->N);
  if (b_need * 4u > a->b_bytes) return DTF_ERR_SHAPE;
  // 16-byte operand loads need 16-byte aligned rows; otherwise take the dword-load build.
  bool vec = (a->K % 4 == 0) && (a->lda % 4 == 0) && dtf_aligned16(a->A);
  if (a->b_transposed) vec = vec && (a->ldb % 4 == 0) && dtf_aligned16(a->B);
  // Deep reductions get more waves per tile so that each wave runs the loop body once or twice.
  if (a->K >= 512) return launch_small_gemm<16>(*a, vec, s);
  if (a->K >= 192) return launch_small_gemm<8>(*a, vec, s);
  if (a->K >= 64) return launch_small_gemm<4>(*a, vec, s);
  return launch_small_gemm<1>(*a, vec, s);
}

// --------------------------------------------------------------------------------------
// grouped weight gradient:  dW = X[rows]^T @ dY,  db = colsum(dY)   (+ fused SGD, + step tail)
// One 16x16 tile per wave, four tiles per workgroup; one extra workgroup runs the step tail.
// --------------------------------------------------------------------------------------
#ifndef DTF_WGRAD_CH
#define DTF_WGRAD_CH 7
#endif

// CH: 16*CH batch rows per pass (row indices, then every operand load, then the MFMAs); 7 covers the reference
// batch of 100 in one pass, 16 quarters the number of dependent passes of a 4096-row batch.
// SPLIT: one tile per workgroup, its batch rows dealt to the four waves in quarters and the partial tiles summed
// through LDS (wave 0 stores).  A 4096-row batch against a [4096, 10] layer is 256 tiles: four per workgroup
// leave three quarters of the CUs idle and every wave with 16 dependent passes; split, all CUs run 4 passes.
template <int CH, bool SPLIT>
__global__ __launch_bounds__(256) void wgrad_grouped_kernel(DtfWgradArgs p) {
  const int nblk_tiles = SPLIT ? p.total_tiles : (p.total_tiles + 3) >> 2;
  if ((int)blockIdx.x >= nblk_tiles) {
    if (p.tail.enable) dtf_tail_body(p.tail, threadIdx.x, blockDim.x, false);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int tile = SPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (!SPLIT && tile >= p.total_tiles) return;
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
  const rsrc_t rx = make_rsrc(q.X, q.x_bytes), ry = make_rsrc(q.dY, q.y_bytes);
  const bool gather = q.x_rows != nullptr;
  const rsrc_t rr = make_rsrc(gather ? (const void*)q.x_rows : (const void*)q.X, gather ? (unsigned)q.Kb * 4u : 0u);
  const unsigned a_lane = (unsigned)am * 4u;
  const unsigned b_lane = (unsigned)bn * 4u;

  // fused update: the old parameter values are requested up front, not after the MFMA chain
  float wold[4] = {0.f, 0.f, 0.f, 0.f};
  float bold = 0.f;
  if (p.mode == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      if (row < q.Mo && bn_ok) wold[r] = q.W[(size_t)row * q.ldw + bn];
    }
    if (tm == 0 && g == 0 && bn_ok) bold = q.b[bn];
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  // SPLIT: wave w takes the passes w, w + 4, ... (k >= Kb reads zeros, so a ragged last pass needs no care)
  for (int kb = SPLIT ? wave * 16 * CH : 0; kb < q.Kb; kb += (SPLIT ? 4 : 1) * 16 * CH) {
    int xr[CH][4];
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + 16 * i + 4 * j + g;
        xr[i][j] = gather ? bldi(rr, (unsigned)k * 4u) : k;   // k >= Kb reads 0 (the descriptor holds Kb entries)
      }
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + 16 * i + 4 * j + g;
        const bool kok = k < q.Kb;
        av[i][j] = bld(rx, (kok && am_ok) ? (unsigned)xr[i][j] * (unsigned)q.ldx * 4u + a_lane : kOOB);
        bv[i][j] = bld(ry, (kok && bn_ok) ? (unsigned)k * (unsigned)q.ldy * 4u + b_lane : kOOB);
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
  if constexpr (SPLIT) {
    __shared__ float part[3][5][64];
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave - 1][r][lane] = acc0[r] + acc1[r];
      part[wave - 1][4][lane] = bsum;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc0[r] = ((acc0[r] + acc1[r]) + part[0][r][lane]) + (part[1][r][lane] + part[2][r][lane]);
    acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    bsum = (bsum + part[0][4][lane]) + (part[1][4][lane] + part[2][4][lane]);
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
    if (q.x_bytes == 0 || q.y_bytes == 0 || q.x_bytes >= kOOB || q.y_bytes >= kOOB) return DTF_ERR_SHAPE;
    if (!q.x_rows && ((uint64_t)(q.Kb - 1) * q.ldx + q.Mo) * 4u > q.x_bytes) return DTF_ERR_SHAPE;
    if (((uint64_t)(q.Kb - 1) * q.ldy + q.N) * 4u > q.y_bytes) return DTF_ERR_SHAPE;
    tiles += q.tiles_n * dtf_cdiv(q.Mo, 16);
  }
  if (tiles != a->total_tiles) return DTF_ERR_SHAPE;
  if (a->tail.enable) {
    if (!a->tail.state) return DTF_ERR_ARG;
    if (a->tail.ridx_next && (a->tail.B <= 0 || a->tail.n_train <= 0 || a->tail.ridx_next == a->tail.ridx_cur)) return DTF_ERR_SHAPE;
    if (a->tail.nparts > 0 && (!a->tail.loss_part || !a->tail.correct_part || !a->tail.post_part)) return DTF_ERR_ARG;
  }
  bool long_k = true;
  for (int i = 0; i < a->nprob; ++i) long_k = long_k && a->prob[i].Kb >= 1024;
  if (long_k) hipLaunchKernelGGL((wgrad_grouped_kernel<16, true>), dim3(tiles + 1), dim3(256), 0, s, *a);
  else hipLaunchKernelGGL((wgrad_grouped_kernel<DTF_WGRAD_CH, false>), dim3(dtf_cdiv(tiles, 4) + 1), dim3(256), 0, s, *a);
  return dtf_check_launch();
}

// --------------------------------------------------------------------------------------
// head: logits, (double) softmax cross-entropy, accuracy, dz2 and dz1.
// One workgroup = 4 waves = 16 rows.  The three dependent phases are each split over the
// four waves so that the serial latency chain per wave stays short:
//   A  z2 partial sums, k-chunks dealt round-robin to the waves (reduced through LDS)
//   B  wave w does the softmax / loss / accuracy / dz2 math of rows 4g + w
//   C  dz1 column chunks dealt round-robin to the waves
// --------------------------------------------------------------------------------------
#ifndef DTF_HEAD_CH
#define DTF_HEAD_CH 2
#endif

// CH = 16-wide k-chunks (phase A) / column chunks (phase C) whose loads a wave has in flight per pass: 2 keeps the
// serial chain of the 100-unit reference layer short, 8 gives a 4096-unit layer 4x fewer dependent passes.
template <int CH>
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
  const bool train_tile = r0 < p.M_train && (p.dz1 != nullptr || p.dz1_bf != nullptr);   // workgroup-uniform
  const int nchunk = (Hd + 15) >> 4;
  const rsrc_t rh = make_rsrc(p.Hact, p.h_bytes), rw = make_rsrc(p.W2, (unsigned)Hd * (unsigned)C * 4u);

  // ---- phase A: z2 = Hact @ W2 (16 rows x 16 padded classes), split-K over the waves ----
  const int arow = r0 + fr;
  const bool arow_ok = arow < p.M_total;
  const unsigned a_row_off = (unsigned)arow * (unsigned)p.ldh * 4u;
  const unsigned b_lane_off = (unsigned)fr * 4u;
  const bool vec = ((p.ldh | Hd) & 3) == 0;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int cb = wave; cb < nchunk; cb += 4 * CH) {
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = ((cb + 4 * i) << 4) + 4 * g;
      if (vec) {
        bld4(rh, (arow_ok && k < Hd) ? a_row_off + (unsigned)k * 4u : kOOB, av[i]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[i][j] = bld(rh, (arow_ok && k + j < Hd) ? a_row_off + (unsigned)(k + j) * 4u : kOOB);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[i][j] = bld(rw, (cls_ok && k + j < Hd) ? (unsigned)(k + j) * (unsigned)C * 4u + b_lane_off : kOOB);
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
  const int lrow = p.rows ? p.rows[row_ok ? row : 0] : row;
  const float labv = p.labels[(size_t)(row_ok ? lrow : 0) * C + (cls_ok ? fr : 0)];
  const float lab = (row_ok && cls_ok) ? labv : 0.f;
  const float biasv = p.b2[cls_ok ? fr : 0];
  __syncthreads();

  // ---- phase B ----
  float loss_acc = 0.f;
  int correct_acc = 0, post_acc = 0;
  {
    const float zsum = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
    const float z = cls_ok ? zsum + biasv : NEG_INF;
    const float zmax = row16_max(z);
    const float e1 = cls_ok ? __expf(z - zmax) : 0.f;
    const float s1 = row16_sum(e1);
    const float y = e1 / s1;
    if (p.probs && row_ok && cls_ok) p.probs[(size_t)row * C + fr] = y;
    float logp, dz;
    if (p.loss_mode == 0) {
      // softmax applied a second time on the probabilities (the reference feeds y as "logits")
      const float ymax = row16_max(cls_ok ? y : NEG_INF);
      const float e2 = cls_ok ? __expf(y - ymax) : 0.f;
      const float s2 = row16_sum(e2);
      const float pp = e2 / s2;
      logp = (y - ymax) - __logf(s2);
      const float dy = (pp - lab) * p.inv_b;
      const float dot = row16_sum(cls_ok ? dy * y : 0.f);
      dz = (dy - dot) * y;
    } else {
      logp = (z - zmax) - __logf(s1);
      dz = (y - lab) * p.inv_b;
    }
    const float lrow_loss = -row16_sum(cls_ok ? lab * logp : 0.f);
    // argmax with lowest-index tie break (tf.argmax, distributed.py:83)
    const int am_y = row16_min((cls_ok && z == zmax) ? fr : 16);
    const float labmax = row16_max(cls_ok ? lab : NEG_INF);
    const int am_l = row16_min((cls_ok && lab == labmax) ? fr : 16);
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
    float cw[CH][4], cact[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int hcol = ((cb + 4 * i) << 4) + fr;
      const bool h_ok = hcol < Hd;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        cw[i][j] = bld(rw, (h_ok && 4 * j + g < C) ? ((unsigned)hcol * (unsigned)C + (unsigned)(4 * j + g)) * 4u : kOOB);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = r0 + 4 * g + r;
        cact[i][r] = bld(rh, (h_ok && orow < p.M_train) ? ((unsigned)orow * (unsigned)p.ldh + (unsigned)hcol) * 4u : kOOB);
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int hcol = ((cb + 4 * i) << 4) + fr;
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = MFMA16(a4[0], cw[i][0], d0);
      d1 = MFMA16(a4[1], cw[i][1], d1);
      d0 = MFMA16(a4[2], cw[i][2], d0);
      d1 = MFMA16(a4[3], cw[i][3], d1);
      float dv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = r0 + 4 * g + r;
        dv[r] = cact[i][r] > 0.f ? d0[r] + d1[r] : 0.f;
        if (p.dz1 && orow < p.M_train && hcol < Hd) p.dz1[(size_t)orow * p.ld_dz1 + hcol] = dv[r];
      }
      if (p.dz1_bf && hcol < Hd) {
        // compute-dtype copies for the bf16 products below: row-major (2 bytes per lane, 32 contiguous bytes per
        // 16 lanes) and transposed (this lane's 4 rows are 8 contiguous bytes)
        unsigned short* bf = static_cast<unsigned short*>(p.dz1_bf);
        unsigned short* bft = static_cast<unsigned short*>(p.dz1_bf_t);
        unsigned short hb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          hb[r] = __builtin_bit_cast(unsigned short, (__bf16)dv[r]);
          if (r0 + 4 * g + r < p.M_train) bf[(size_t)(r0 + 4 * g + r) * p.ld_bf + hcol] = hb[r];
        }
        if (bft) {
          const size_t e = (size_t)hcol * p.ld_bf_t + r0 + 4 * g;
          if (r0 + 4 * g + 3 < p.M_train) {
            uint2 q;
            q.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
            q.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
            *reinterpret_cast<uint2*>(bft + e) = q;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r0 + 4 * g + r < p.M_train) bft[e + r] = hb[r];
          }
        }
      }
    }
  }
}

DTF_EXPORT int dtf_head(const DtfHeadArgs* a, hipStream_t s) {
  if (!a || !a->Hact || !a->W2 || !a->b2 || !a->labels) return DTF_ERR_ARG;
  if (a->M_total <= 0 || a->M_train < 0 || a->M_train > a->M_total) return DTF_ERR_SHAPE;
  if (a->C <= 0 || a->C > 16 || a->Hd <= 0 || a->ldh < a->Hd) return DTF_ERR_SHAPE;
  if (a->M_train > 0 && a->dz1 && a->ld_dz1 < a->Hd) return DTF_ERR_SHAPE;
  if (a->dz1_bf_t && !a->dz1_bf) return DTF_ERR_ARG;
  if (a->M_train > 0 && a->dz1_bf && a->ld_bf < a->Hd) return DTF_ERR_SHAPE;
  if (a->M_train > 0 && a->dz1_bf_t && (a->ld_bf_t < a->M_train || (a->ld_bf_t & 3) || (reinterpret_cast<uintptr_t>(a->dz1_bf_t) & 7u)))
    return DTF_ERR_ALIGN;
  if (((a->ldh | a->Hd) & 3) == 0 && !dtf_aligned16(a->Hact)) return DTF_ERR_ALIGN;
  if (a->h_bytes == 0 || a->h_bytes >= kOOB || ((uint64_t)(a->M_total - 1) * a->ldh + a->Hd) * 4u > a->h_bytes) return DTF_ERR_SHAPE;
  if ((uint64_t)a->Hd * a->C * 4u >= kOOB) return DTF_ERR_SHAPE;
  if (a->Hd >= 1024) hipLaunchKernelGGL(head_kernel<8>, dim3(dtf_cdiv(a->M_total, 16)), dim3(256), 0, s, *a);
  else hipLaunchKernelGGL(head_kernel<DTF_HEAD_CH>, dim3(dtf_cdiv(a->M_total, 16)), dim3(256), 0, s, *a);
  return dtf_check_launch();
}
