// Two-launch training step of the reference network x -> relu(x W1 + b1) -> softmax(h W2 + b2)
// (math: /root/reference/distributed.py:78-89 and its autodiff), exact fp32 MFMA throughout.
//
// The three-launch step (mlp_small.hip) pays two kernel boundaries (~1.9 us each on this chip) plus
// two store -> load round trips through L2 between the forward pass and the weight gradients.
// Here everything after the hidden layer runs in ONE launch, which is possible without any
// inter-workgroup synchronisation because the softmax gradient is cheap enough to recompute:
//
//   launch 1 (fwd):  H = relu(X[rows] W1 + b1), one 16x16 tile per 16-wave workgroup (split-K).
//                    The finished tile is multiplied by its 16 rows of W2 right away and stored as a
//                    partial-logit slab zpart[row][class][column tile]; b2 rides on tile 0.  The
//                    launch also snapshots W2 and the step counters for launch 2.
//   launch 2 (bwd):  - R/16 "stat" workgroups: logits = sum of slabs, loss / accuracy partials of their
//                      16 rows, written straight into the step's wide log record (summed by the host).
//                    - ceil(tiles/4) "train" workgroups, one weight-gradient tile per wave.  Every
//                      workgroup first recomputes dz2 for the whole batch from the slabs (7 rows of
//                      math per lane) into LDS; a dW1 wave then rebuilds its dz1 tile with 3-4 MFMAs
//                      per 16 batch rows, and because the MFMA result layout (reg r = row 4g+r,
//                      col = lane&15) is exactly the B-operand layout of the next MFMA when batch
//                      row 16rb+4g+j is fed to instruction j, dz1 never leaves registers.
//                    - one tail workgroup: counters, next batch rows.
//   Launch 2 updates W2/b2 while other workgroups still need W2 for dz1: they read the snapshot.
#include "dtf_device.h"

// Diagnostic build only (-DDTF_STAMPS): per-workgroup timestamps of the 100 MHz constant clock.
#ifdef DTF_STAMPS
__device__ unsigned long long* mlp2_stamp_buf = nullptr;
#define M2_STAMP(slot, wg)                                                                                                \
  do {                                                                                                                    \
    if (threadIdx.x == 0 && mlp2_stamp_buf) mlp2_stamp_buf[(size_t)(wg) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int dtf_set_stamp_buffer_mlp2(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mlp2_stamp_buf), &p, sizeof(p));
}
#else
#define M2_STAMP(slot, wg) do { } while (0)
#endif

__device__ __forceinline__ void bld4i(rsrc_t r, unsigned off, int (&v)[4]) {
  auto t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  v[0] = (int)t[0]; v[1] = (int)t[1]; v[2] = (int)t[2]; v[3] = (int)t[3];
}
// position of class c inside a 16-float LDS row: a 16-byte read at 4*g returns classes g, 4+g, 8+g, 12+g
__device__ __forceinline__ int dz_slot(int c) { return 4 * (c & 3) + (c >> 2); }

// ------------------------------------------------------------------------------------------
// launch 1
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mlp2_fwd_kernel(DtfMlp2Args p) {
  __shared__ float red[16][4][64];
  __shared__ float hs[16][17];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int nt = blockIdx.x, n0 = nt * 16, m0 = blockIdx.y * 16;
  const int Hd = p.Hd, C = p.C;
  const rsrc_t ra = make_rsrc(p.X, p.x_bytes), rb = make_rsrc(p.W1, (unsigned)p.In * (unsigned)Hd * 4u);
  const int arow = m0 + fr;
  const int asrc = p.ridx[arow];          // the table is padded to whole row blocks
  const unsigned a_row_off = arow < p.R ? (unsigned)asrc * (unsigned)p.ldx * 4u : kOOB;
  const int bcol = n0 + fr;
  const unsigned b_lane_off = bcol < Hd ? (unsigned)bcol * 4u : kOOB;

  const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
  const int e_rl = 4 * (e_ln >> 4) + e_r, e_cl = e_ln & 15;
  const int e_row = m0 + e_rl, e_col = n0 + e_cl;
  const bool e_ok = threadIdx.x < 256 && e_row < p.R && e_col < Hd;
  const float e_bias = e_ok ? p.b1[e_col] : 0.f;

  float w2v[4] = {0.f, 0.f, 0.f, 0.f};
  float b2v = 0.f;
  if (wave == 0) {
    const rsrc_t rw2 = make_rsrc(p.W2, (unsigned)Hd * (unsigned)C * 4u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int hcol = n0 + 4 * j + g;
      w2v[j] = bld(rw2, (hcol < Hd && fr < C) ? ((unsigned)hcol * (unsigned)C + (unsigned)fr) * 4u : kOOB);
    }
    if (nt == 0 && fr < C) b2v = p.b2[fr];
  }
  int snap_idx = -1;
  float snapv = 0.f;
  if (blockIdx.y == 0 && (int)threadIdx.x < 16 * C) {
    const int idx = n0 * C + (int)threadIdx.x;
    if (idx < Hd * C) { snap_idx = idx; snapv = p.W2[idx]; }
  }
  // the last column tile of every row block also gathers the label rows of its 16 batch rows
  float labv = 0.f;
  const bool lab_wg = blockIdx.x == gridDim.x - 1 && threadIdx.x < 256;
  if (lab_wg) {
    const int row = m0 + ((int)threadIdx.x >> 4), c = (int)threadIdx.x & 15;
    const bool ok = row < p.R && c < C;
    const float v = p.labels[ok ? (size_t)p.ridx[row] * C + c : 0];
    labv = ok ? v : 0.f;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) p.ctr_snap[threadIdx.x] = p.tail.state[threadIdx.x];

  M2_STAMP(0, blockIdx.y * gridDim.x + blockIdx.x);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nchunk = (p.In + 15) >> 4;
  constexpr int CH = 4;
  for (int cb = wave; cb < nchunk; cb += 16 * CH) {
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = ((cb + i * 16) << 4) + 4 * g;
      bld4(ra, (a_row_off != kOOB && k < p.In) ? a_row_off + (unsigned)k * 4u : kOOB, av[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bv[i][j] = bld(rb, (b_lane_off != kOOB && k + j < p.In) ? b_lane_off + (unsigned)(k + j) * (unsigned)Hd * 4u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      acc0 = MFMA16(av[i][0], bv[i][0], acc0);
      acc1 = MFMA16(av[i][1], bv[i][1], acc1);
      acc0 = MFMA16(av[i][2], bv[i][2], acc0);
      acc1 = MFMA16(av[i][3], bv[i][3], acc1);
    }
  }
  M2_STAMP(1, blockIdx.y * gridDim.x + blockIdx.x);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  __syncthreads();
  M2_STAMP(2, blockIdx.y * gridDim.x + blockIdx.x);
  if (threadIdx.x < 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][e_r][e_ln];
    s = fmaxf(s + e_bias, 0.f);
    if (e_ok) p.H[(size_t)e_row * p.ldh + e_col] = s;
    hs[e_rl][e_cl] = e_ok ? s : 0.f;
  }
  __syncthreads();
  M2_STAMP(3, blockIdx.y * gridDim.x + blockIdx.x);
  if (wave == 0) {
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
    z0 = MFMA16(hs[fr][g], w2v[0], z0);
    z1 = MFMA16(hs[fr][4 + g], w2v[1], z1);
    z0 = MFMA16(hs[fr][8 + g], w2v[2], z0);
    z1 = MFMA16(hs[fr][12 + g], w2v[3], z1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      if (row < p.R) p.zpart[((size_t)row * 16 + fr) * 8 + nt] = (z0[r] + z1[r]) + b2v;
    }
  }
  if (snap_idx >= 0) p.W2s[snap_idx] = snapv;
  if (lab_wg) p.labg[(size_t)m0 * 16 + threadIdx.x] = labv;
  M2_STAMP(4, blockIdx.y * gridDim.x + blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// launch 2
// ------------------------------------------------------------------------------------------
// Softmax / loss math of one row spread over 16 lanes (lane & 15 = class).  Returns d loss / d logit.
template <bool FULL>
__device__ __forceinline__ float mlp2_row_math(float zsum, float lab, bool cls_ok, int fr, int loss_mode, float inv_b,
                                               float& loss, int& hit) {
  const float NEG_INF = -__builtin_huge_valf();
  const float z = cls_ok ? zsum : NEG_INF;
  const float zmax = row16_max(z);
  const float e1 = __expf(z - zmax);                 // 0 on the padding lanes
  const float s1 = row16_sum(e1);
  const float y = e1 * __builtin_amdgcn_rcpf(s1);
  float logp, dz;
  if (loss_mode == 0) {
    // softmax applied a second time on the probabilities (the reference feeds y as "logits")
    const float ymax = row16_max(cls_ok ? y : NEG_INF);
    const float e2 = cls_ok ? __expf(y - ymax) : 0.f;
    const float s2 = row16_sum(e2);
    const float pp = e2 * __builtin_amdgcn_rcpf(s2);
    const float dy = (pp - lab) * inv_b;
    const float dot = row16_sum(dy * y);
    dz = (dy - dot) * y;
    if (FULL) logp = (y - ymax) - __logf(s2);
  } else {
    dz = (y - lab) * inv_b;
    if (FULL) logp = (z - zmax) - __logf(s1);
  }
  if (FULL) {
    loss = -row16_sum(cls_ok ? lab * logp : 0.f);
    // argmax with lowest-index tie break (tf.argmax, distributed.py:83)
    const int am_y = row16_min((cls_ok && z == zmax) ? fr : 16);
    const float labmax = row16_max(cls_ok ? lab : NEG_INF);
    const int am_l = row16_min((cls_ok && lab == labmax) ? fr : 16);
    hit = am_y == am_l ? 1 : 0;
  }
  return dz;
}

__device__ __forceinline__ float sum8(const float (&a)[4], const float (&b)[4]) {
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
}

__device__ __forceinline__ void mlp2_stat_block(const DtfMlp2Args& p, int rb, unsigned zbytes) {
  __shared__ float part_loss[4];
  __shared__ int part_cnt[4][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int C = p.C;
  const bool cls_ok = fr < C;
  const int row = 16 * rb + 4 * g + wave;
  const bool row_ok = row < p.R, train_row = row < p.B;
  M2_STAMP(0, blockIdx.x);
  const rsrc_t rz = make_rsrc(p.zpart, zbytes), rl = make_rsrc(p.labg, zbytes >> 3);
  float za[4], zb[4];
  bld4(rz, (unsigned)(row * 16 + fr) * 32u, za);
  bld4(rz, (unsigned)(row * 16 + fr) * 32u + 16u, zb);
  const float lab = bld(rl, (unsigned)(row * 16 + fr) * 4u);     // 0 on padding rows / classes
  const int ls = p.ctr_snap[1];
  float loss = 0.f;
  int hit = 0;
  (void)mlp2_row_math<true>(sum8(za, zb), lab, cls_ok, fr, p.loss_mode, p.inv_b, loss, hit);
  float loss_acc = 0.f;
  int correct_acc = 0, post_acc = 0;
  if (fr == 0 && row_ok) {
    if (train_row) { loss_acc = loss; correct_acc = hit; }
    else post_acc = hit;
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
    int* rec = p.wlog + (size_t)DTF_WLOG_INTS * (ls % p.log_cap);
    rec[4 + rb] = __float_as_int((part_loss[0] + part_loss[1]) + (part_loss[2] + part_loss[3]));
    rec[20 + rb] = part_cnt[0][0] + part_cnt[1][0] + part_cnt[2][0] + part_cnt[3][0];
    rec[36 + rb] = part_cnt[0][1] + part_cnt[1][1] + part_cnt[2][1] + part_cnt[3][1];
  }
  M2_STAMP(6, blockIdx.x);
}

__device__ __forceinline__ void mlp2_tail_block(const DtfMlp2Args& p) {
  const DtfTailArgs& t = p.tail;
  const int tid = threadIdx.x;
  M2_STAMP(0, blockIdx.x);
  const int gs = t.state[0], ls = t.state[1];
  __syncthreads();   // every thread holds the counters before thread 0 advances them
  if (tid == 0) {
    int* rec = p.wlog + (size_t)DTF_WLOG_INTS * (ls % p.log_cap);
    rec[0] = gs;       // value of global_step before this step's increment
    rec[1] = ls;
    t.state[0] = gs + t.gs_inc;
    t.state[1] = ls + 1;
  }
  if (t.ridx_next) {
    const int next = ls + 1;
    for (int i = tid; i < t.B; i += 256) {
      const long long q = (long long)next * t.B + i;
      const long long epoch = q / t.n_train;
      const int pos = (int)(q - epoch * t.n_train);
      const int row = t.perm ? t.perm[(size_t)(epoch & 1) * t.n_train + pos] : pos;
      t.ridx_next[t.B + i] = t.ridx_cur[i];   // the batch just trained on becomes the accuracy-only rows
      t.ridx_next[i] = row;
    }
  }
  M2_STAMP(6, blockIdx.x);
}

// One weight-gradient tile per wave.  IS2 = false: tiles of dW1 (inputs x hidden), dz1 rebuilt in registers;
// IS2 = true: tiles of dW2 (hidden x classes).  `tile0` is the first tile of this workgroup within its kind.
template <int NJ, int NRB, bool IS2>
__device__ __forceinline__ void mlp2_train_block(const DtfMlp2Args& p, int tile0, unsigned zbytes, float* dzs) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int C = p.C, Hd = p.Hd, B = p.B;
  const int NT = (Hd + 15) >> 4;
  const int ntiles = IS2 ? NT : ((p.In + 15) >> 4) * NT;
  const int tile = tile0 + wave;                    // wave-uniform
  const bool has_tile = tile < ntiles;
  const int mt = IS2 ? tile : tile / NT;
  const int nt = IS2 ? 0 : tile - mt * NT;
  const int m0 = mt * 16, n0 = nt * 16;
  const int Mo = IS2 ? Hd : p.In, N = IS2 ? C : Hd;
  const bool am_ok = has_tile && m0 + fr < Mo, bn_ok = has_tile && n0 + fr < N;
  const bool cls_ok = fr < C;
  const unsigned h_bytes = (unsigned)(((size_t)(p.R - 1) * p.ldh + Hd) * 4u);
  M2_STAMP(0, blockIdx.x);

  // ---- operands of the softmax-gradient pass: this wave owns rows 16*rb + 4*g + wave ----
  const rsrc_t rz = make_rsrc(p.zpart, zbytes), rl = make_rsrc(p.labg, zbytes >> 3);
  float za[NRB][4], zb[NRB][4], lab[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const unsigned e = (unsigned)((16 * rb + 4 * g + wave) * 16 + fr);   // past the buffers: reads 0
    bld4(rz, e * 32u, za[rb]);
    bld4(rz, e * 32u + 16u, zb[rb]);
    lab[rb] = bld(rl, e * 4u);
  }
  // ---- operands of the tile, requested now so that they arrive during the softmax pass ----
  float* const Wt = IS2 ? p.W2 : p.W1;
  float* const bt = IS2 ? p.b2 : p.b1;
  const rsrc_t rwt = make_rsrc(Wt, (unsigned)Mo * (unsigned)N * 4u), rbt = make_rsrc(bt, (unsigned)N * 4u);
  const rsrc_t rh = make_rsrc(p.H, h_bytes);
  int xr[NRB][4];
  float av[NRB][4], hm[NRB][4], wb[4] = {0.f, 0.f, 0.f, 0.f}, wold[4] = {0.f, 0.f, 0.f, 0.f}, bold = 0.f;
  if (!IS2) {
    const rsrc_t rri = make_rsrc(p.ridx, (unsigned)p.ridx_len * 4u);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) bld4i(rri, (unsigned)(16 * rb + 4 * g) * 4u, xr[rb]);
    const rsrc_t rw = make_rsrc(p.W2s, (unsigned)Hd * (unsigned)C * 4u);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rb + 4 * g + r;
        hm[rb][r] = bld(rh, (bn_ok && row < B) ? ((unsigned)row * (unsigned)p.ldh + (unsigned)(n0 + fr)) * 4u : kOOB);
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      wb[j] = bld(rw, (bn_ok && 4 * j + g < C) ? ((unsigned)(n0 + fr) * (unsigned)C + (unsigned)(4 * j + g)) * 4u : kOOB);
  }
  if (p.mode == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      wold[r] = bld(rwt, (row < Mo && bn_ok) ? ((unsigned)row * (unsigned)N + (unsigned)(n0 + fr)) * 4u : kOOB);
    }
    bold = bld(rbt, (mt == 0 && g == 0 && bn_ok) ? (unsigned)(n0 + fr) * 4u : kOOB);
  }
  {
    const rsrc_t ra = IS2 ? rh : make_rsrc(p.X, p.x_bytes);
    const unsigned lda4 = (unsigned)(IS2 ? p.ldh : p.ldx) * 4u;
    const unsigned a_lane = (unsigned)(m0 + fr) * 4u;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 16 * rb + 4 * g + j;                      // batch row fed to MFMA j of row block rb
        const unsigned src = IS2 ? (unsigned)k : (unsigned)xr[rb][j];
        av[rb][j] = bld(ra, (am_ok && k < B) ? src * lda4 + a_lane : kOOB);
      }
  }
  M2_STAMP(1, blockIdx.x);

  // ---- softmax gradient of the whole batch into LDS (recomputed by every workgroup); the row blocks are
  //      independent dependency chains, left unrolled so that they interleave ----
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const int row = 16 * rb + 4 * g + wave;
    float loss;
    int hit;
    const float dz = mlp2_row_math<false>(sum8(za[rb], zb[rb]), lab[rb], cls_ok, fr, p.loss_mode, p.inv_b, loss, hit);
    dzs[row * 20 + dz_slot(fr)] = (row < B && cls_ok) ? dz : 0.f;
  }
  M2_STAMP(3, blockIdx.x);
  __syncthreads();
  M2_STAMP(4, blockIdx.x);
  if (!has_tile) return;

  // ---- the tile ----
  float bq[NRB][4];
  if (!IS2) {
    // dz1 tile = (dz2[16 rows] W2^T[:, 16 columns]) masked by the ReLU; classes ride the K dimension
    f32x4 a4[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) a4[rb] = *reinterpret_cast<const f32x4*>(&dzs[(16 * rb + fr) * 20 + 4 * g]);
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = MFMA16(a4[rb][0], wb[0], d0);
      if (NJ > 1) d1 = MFMA16(a4[rb][1], wb[1], d1);
      if (NJ > 2) d0 = MFMA16(a4[rb][2], wb[2], d0);
      if (NJ > 3) d1 = MFMA16(a4[rb][3], wb[3], d1);
#pragma unroll
      for (int r = 0; r < 4; ++r) bq[rb][r] = hm[rb][r] > 0.f ? d0[r] + d1[r] : 0.f;
    }
  } else {
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[rb][j] = dzs[(16 * rb + 4 * g + j) * 20 + dz_slot(fr)];
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    bsum += (bq[rb][0] + bq[rb][1]) + (bq[rb][2] + bq[rb][3]);
    acc0 = MFMA16(av[rb][0], bq[rb][0], acc0);
    acc1 = MFMA16(av[rb][1], bq[rb][1], acc1);
    acc0 = MFMA16(av[rb][2], bq[rb][2], acc0);
    acc1 = MFMA16(av[rb][3], bq[rb][3], acc1);
  }
  M2_STAMP(5, blockIdx.x);
  float* const dWt = IS2 ? p.dW2 : p.dW1;
  float* const dbt = IS2 ? p.db2 : p.db1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * g + r;
    if (row < Mo && bn_ok) {
      const float gval = acc0[r] + acc1[r];
      const size_t off = (size_t)row * N + n0 + fr;
      if (p.mode == 1) Wt[off] = wold[r] - p.lr * gval;
      else dWt[off] = gval;
    }
  }
  if (mt == 0) {  // wave-uniform: this tile also owns the bias gradient of its 16 columns
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (g == 0 && bn_ok) {
      if (p.mode == 1) bt[n0 + fr] = bold - p.lr * bsum;
      else dbt[n0 + fr] = bsum;
    }
  }
  M2_STAMP(6, blockIdx.x);
}

// workgroups of launch 2: [0, R16) stat | nb1 blocks of dW1 tiles | nb2 blocks of dW2 tiles | 1 tail
template <int NJ, int NRB>
__global__ __launch_bounds__(256) void mlp2_bwd_kernel(DtfMlp2Args p) {
  __shared__ float dzs[128 * 20];
  const int R16 = (p.R + 15) >> 4;
  const int NT = (p.Hd + 15) >> 4;
  const int nb1 = (((p.In + 15) >> 4) * NT + 3) >> 2, nb2 = (NT + 3) >> 2;
  const unsigned zbytes = (unsigned)R16 * 16u * 16u * 8u * 4u;
  const int b = blockIdx.x;
  if (b < R16) mlp2_stat_block(p, b, zbytes);
  else if (b < R16 + nb1) mlp2_train_block<NJ, NRB, false>(p, (b - R16) * 4, zbytes, dzs);
  else if (b < R16 + nb1 + nb2) mlp2_train_block<NJ, NRB, true>(p, (b - R16 - nb1) * 4, zbytes, dzs);
  else mlp2_tail_block(p);
}

static int mlp2_validate(const DtfMlp2Args* a) {
  if (!a || !a->X || !a->ridx || !a->labels || !a->H || !a->zpart || !a->labg || !a->W2s || !a->ctr_snap || !a->wlog) return DTF_ERR_ARG;
  if (!a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->tail.state) return DTF_ERR_ARG;
  if (a->mode == 0 ? (!a->dW1 || !a->db1 || !a->dW2 || !a->db2) : a->mode != 1) return DTF_ERR_ARG;
  if (a->In <= 0 || a->Hd <= 0 || a->Hd > 128 || a->C <= 0 || a->C > 16 || a->B <= 0 || a->B > 128) return DTF_ERR_SHAPE;
  if (a->R != a->B && a->R != 2 * a->B) return DTF_ERR_SHAPE;
  if (a->ldx < a->In || a->ldh < a->Hd || a->log_cap <= 0) return DTF_ERR_SHAPE;
  if ((a->In & 3) || (a->ldx & 3)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->X) || !dtf_aligned16(a->ridx) || !dtf_aligned16(a->zpart)) return DTF_ERR_ALIGN;
  // every row index up to the padded row count is read unguarded
  if ((a->ridx_len & 15) || a->ridx_len < 16 * dtf_cdiv(a->R, 16) || a->ridx_len < 128) return DTF_ERR_SHAPE;
  if (a->x_bytes == 0 || a->x_bytes >= 0xFFFFFFF0u) return DTF_ERR_SHAPE;
  if (a->tail.enable) {
    if (a->tail.B != a->B || a->tail.n_train <= 0 || a->tail.log_cap != a->log_cap) return DTF_ERR_SHAPE;
    if (a->tail.ridx_next && (!a->tail.ridx_cur || a->tail.ridx_next == a->tail.ridx_cur)) return DTF_ERR_SHAPE;
  }
  return DTF_OK;
}

DTF_EXPORT int dtf_mlp2_fwd(const DtfMlp2Args* a, hipStream_t s) {
  const int rc = mlp2_validate(a);
  if (rc != DTF_OK) return rc;
  hipLaunchKernelGGL(mlp2_fwd_kernel, dim3(dtf_cdiv(a->Hd, 16), dtf_cdiv(a->R, 16)), dim3(1024), 0, s, *a);
  return dtf_check_launch();
}

template <int NJ>
static void mlp2_launch_bwd(const DtfMlp2Args& a, int nblk, hipStream_t s) {
  const int nrb = dtf_cdiv(a.B, 16);
  if (nrb <= 2) hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 2>), dim3(nblk), dim3(256), 0, s, a);
  else if (nrb <= 4) hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 4>), dim3(nblk), dim3(256), 0, s, a);
  else if (nrb <= 7) hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 7>), dim3(nblk), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 8>), dim3(nblk), dim3(256), 0, s, a);
}

DTF_EXPORT int dtf_mlp2_bwd(const DtfMlp2Args* a, hipStream_t s) {
  const int rc = mlp2_validate(a);
  if (rc != DTF_OK) return rc;
  if (!a->tail.enable) return DTF_ERR_ARG;
  const int NT = dtf_cdiv(a->Hd, 16);
  const int nblk = dtf_cdiv(a->R, 16) + dtf_cdiv(dtf_cdiv(a->In, 16) * NT, 4) + dtf_cdiv(NT, 4) + 1;
  if (a->C <= 12) mlp2_launch_bwd<3>(*a, nblk, s);
  else mlp2_launch_bwd<4>(*a, nblk, s);
  return dtf_check_launch();
}
