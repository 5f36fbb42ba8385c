// Two-launch training step of the reference network x -> relu(x W1 + b1) -> softmax(h W2 + b2)
// (math: /root/reference/distributed.py:78-89 and its autodiff), exact fp32 MFMA throughout.
//
// What bounds this step is not arithmetic but (a) kernel boundaries (~1.9 us each) and (b) dependent
// memory hops: L2 is written back / invalidated at every kernel boundary, so the first touch of anything
// the previous launch produced costs a trip to memory (~1 us), and an index -> row gather costs two.
// The design therefore is: two launches, and no load in either of them depends on another load.
//
//   launch 1 (fwd):  H = relu(Xg W1 + b1), one 16x16 tile per 16-wave workgroup (split-K).  Xg holds the
//                    batch rows already gathered (by the previous step's launch 2).  The finished tile is
//                    multiplied by its 16 rows of W2 right away and stored as a partial-logit slab
//                    zpart[class][row][column tile]; b2 rides on tile 0.  The launch also snapshots W2 and
//                    the step counters for launch 2.
//   launch 2 (bwd):  - R/16 "stat" workgroups: logits = sum of slabs, loss / accuracy partials of their
//                      16 rows, written straight into the step's wide log record (summed by the host).
//                    - "train" workgroups, one weight-gradient tile per wave.  Every workgroup first
//                      recomputes dz2 of the whole batch into LDS: one batch row per lane, the classes in
//                      registers (no cross-lane reductions).  A dW1 wave then rebuilds its dz1 tile with
//                      3-4 MFMAs per 16 batch rows; the MFMA result layout (reg r = row 4g+r,
//                      col = lane&15) is exactly the B-operand layout of the next MFMA when batch row
//                      16rb+4g+j is fed to instruction j, so dz1 never leaves registers.
//                    - "tail" workgroups: counters, and the NEXT step's batch: row indices, gathered
//                      image rows and label rows, off the critical path of the train workgroups.
//   Launch 2 updates W2/b2 while other workgroups still need W2 for dz1: they read the snapshot.
//   Per-step buffers (Xg, labg, ridx) rotate with period 3: launch 2 of step s reads slot s (batch) and
//   s-1 (accuracy-only rows = the previous batch) while its tail workgroups fill slot s+1.
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

// raw buffer loads with a wave-uniform (scalar) offset added to the per-lane one: the per-load address
// arithmetic then runs on the scalar unit instead of costing several VALU operations per load
__device__ __forceinline__ float bld_s(rsrc_t r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bld4_s(rsrc_t r, unsigned voff, unsigned soff, float (&v)[4]) {
  auto t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
}

#define MLP2_ROWS 128               // rows per slot of Xg / labg (batch rows beyond B stay zero)
#define MLP2_LABG_ROWS 128          // batch rows per class in a labg slot
#define MLP2_LABG_SLOT (16 * MLP2_LABG_ROWS)

__device__ __forceinline__ int mlp2_prev_slot(int q) { return q == 0 ? 2 : q - 1; }
__device__ __forceinline__ int mlp2_next_slot(int q) { return q == 2 ? 0 : q + 1; }

// ------------------------------------------------------------------------------------------
// launch 1
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mlp2_fwd_kernel(DtfMlp2Args p) {
  __shared__ float red[16][4][64];
  __shared__ float hs[16][17];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  // XCD-major block index, row blocks outermost: the workgroups of an XCD share batch rows (see mlp2_bwd_kernel)
  const int lin = dtf_xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int nt = lin % (int)gridDim.x, by = lin / (int)gridDim.x;
  const int n0 = nt * 16, m0 = by * 16;
  const int Hd = p.Hd, C = p.C, B = p.B;
  const int Rp = ((p.R + 15) >> 4) << 4;
  const unsigned slot_bytes = (unsigned)MLP2_ROWS * (unsigned)p.In * 4u;
  const rsrc_t ra = make_rsrc(p.Xg, 3u * slot_bytes), rb = make_rsrc(p.W1, (unsigned)p.In * (unsigned)Hd * 4u);
  // logical row r: [0,B) = this step's batch (slot q), [B,R) = the previous batch (slot q-1)
  const int arow = m0 + fr;
  const int aslot = arow < B ? p.q : mlp2_prev_slot(p.q);
  const int asrc = arow < B ? arow : arow - B;
  const unsigned a_row_off = arow < p.R ? (unsigned)aslot * slot_bytes + (unsigned)asrc * (unsigned)p.In * 4u : kOOB;
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
  if (by == 0 && (int)threadIdx.x < 16 * C) {
    const int idx = n0 * C + (int)threadIdx.x;
    if (idx < Hd * C) { snap_idx = idx; snapv = p.W2[idx]; }
  }
  if (nt == 0 && by == 0 && threadIdx.x < 2) p.ctr_snap[threadIdx.x] = p.tail.state[threadIdx.x];
  // asynchronous push with Adam: how many updates every slice has received, read once for the whole step (the tail
  // of launch 2 advances the counters while the other workgroups of that launch are still computing their updates)
  if (p.mode == 3 && p.async_.optimizer == 1 && nt == 0 && by == 0 && threadIdx.x >= 64 &&
      threadIdx.x < 64 + (unsigned)p.async_.shards)
    p.async_.applied_snap[threadIdx.x - 64] =
        __hip_atomic_load(p.async_.counters[threadIdx.x - 64] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

  M2_STAMP(0, lin);
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
  M2_STAMP(1, lin);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  __syncthreads();
  M2_STAMP(2, lin);
  if (threadIdx.x < 256) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][e_r][e_ln];
    s = fmaxf(s + e_bias, 0.f);
    if (e_ok) p.H[(size_t)e_row * p.ldh + e_col] = s;
    hs[e_rl][e_cl] = e_ok ? s : 0.f;
  }
  __syncthreads();
  M2_STAMP(3, lin);
  if (wave == 0) {
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
    z0 = MFMA16(hs[fr][g], w2v[0], z0);
    z1 = MFMA16(hs[fr][4 + g], w2v[1], z1);
    z0 = MFMA16(hs[fr][8 + g], w2v[2], z0);
    z1 = MFMA16(hs[fr][12 + g], w2v[3], z1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      if (row < p.R) p.zpart[((size_t)fr * Rp + row) * 8 + nt] = (z0[r] + z1[r]) + b2v;
    }
  }
  if (snap_idx >= 0) p.W2s[snap_idx] = snapv;
  M2_STAMP(4, lin);
}

// ------------------------------------------------------------------------------------------
// launch 2
// ------------------------------------------------------------------------------------------
// Softmax / loss math of one row spread over 16 lanes (lane & 15 = class): the stat workgroups.
__device__ __forceinline__ void mlp2_row_stats(float zsum, float lab, bool cls_ok, int fr, int loss_mode, float& loss,
                                               int& hit) {
  const float NEG_INF = -__builtin_huge_valf();
  const float z = cls_ok ? zsum : NEG_INF;
  const float zmax = row16_max(z);
  const float e1 = __expf(z - zmax);                 // 0 on the padding lanes
  const float s1 = row16_sum(e1);
  float logp;
  if (loss_mode == 0) {
    // softmax applied a second time on the probabilities (the reference feeds y as "logits")
    const float y = e1 * __builtin_amdgcn_rcpf(s1);
    const float ymax = row16_max(cls_ok ? y : NEG_INF);
    const float e2 = cls_ok ? __expf(y - ymax) : 0.f;
    const float s2 = row16_sum(e2);
    logp = (y - ymax) - __logf(s2);
  } else {
    logp = (z - zmax) - __logf(s1);
  }
  loss = -row16_sum(cls_ok ? lab * logp : 0.f);
  // argmax with lowest-index tie break (tf.argmax, distributed.py:83)
  const int am_y = row16_min((cls_ok && z == zmax) ? fr : 16);
  const float labmax = row16_max(cls_ok ? lab : NEG_INF);
  const int am_l = row16_min((cls_ok && lab == labmax) ? fr : 16);
  hit = am_y == am_l ? 1 : 0;
}

__device__ __forceinline__ float sum8(const float (&a)[4], const float (&b)[4]) {
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
}

__device__ __forceinline__ void mlp2_stat_block(const DtfMlp2Args& p, int rb, int Rp, unsigned zbytes) {
  __shared__ float part_loss[4];
  __shared__ int part_cnt[4][2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const bool cls_ok = fr < p.C;
  const int row = 16 * rb + 4 * g + wave;
  const bool row_ok = row < p.R, train_row = row < p.B;
  M2_STAMP(0, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
  const rsrc_t rz = make_rsrc(p.zpart, zbytes), rl = make_rsrc(p.labg, 3u * MLP2_LABG_SLOT * 4u);
  float za[4], zb[4];
  const unsigned zoff = (unsigned)(fr * Rp + row) * 32u;
  bld4(rz, zoff, za);
  bld4(rz, zoff + 16u, zb);
  const int lslot = train_row ? p.q : mlp2_prev_slot(p.q);
  const int lrow = train_row ? row : row - p.B;
  const float lab = bld(rl, row_ok ? (unsigned)(lslot * MLP2_LABG_SLOT + fr * MLP2_LABG_ROWS + lrow) * 4u : kOOB);
  const int ls = p.ctr_snap[1];
  float loss = 0.f;
  int hit = 0;
  mlp2_row_stats(sum8(za, zb), lab, cls_ok, fr, p.loss_mode, loss, hit);
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
  M2_STAMP(6, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
}

// Asynchronous push inside the backward launch: does the step that makes `ls1` steps of this worker take the master's
// values as its replica (the pull)?  The same answer in every workgroup of the launch: the forced-pull word of this
// step's parity was written by the previous launch.
__device__ __forceinline__ bool mlp2_async_pulls(const DtfMlp2Async& as, int ls1) {
  if (as.pull_every <= 1 || (ls1 % as.pull_every) == 0) return true;
  return as.bound >= 0 && as.astate && as.astate[6 + (ls1 & 1)] != 0;
}

// One element of the ps-resident Adam update (the arithmetic of async_push_pull_kernel): `applied` = updates the
// element's slice had received when this worker looked.
__device__ __forceinline__ float mlp2_async_adam(const DtfMlp2Async& as, float lr, float w, float g, float& m, float& v, int applied) {
  const float t = (float)(applied + 1);
  const float lr_t = lr * sqrtf(1.f - powf(as.b2, t)) / (1.f - powf(as.b1, t));
  m = as.b1 * m + (1.f - as.b1) * g;
  v = as.b2 * v + (1.f - as.b2) * (g * g);
  return w - lr_t * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + as.eps);
}

// Tail workgroup tb: counters (tb == 0) and rows [8*tb, 8*tb+8) of the next step's batch: 32 threads per row
// copy the image row into the next Xg slot, the label row into the next labg slot, the index into ridx.
template <int MODE>
__device__ __forceinline__ void mlp2_tail_block(const DtfMlp2Args& p, int tb) {
  const DtfTailArgs& t = p.tail;
  const int tid = threadIdx.x;
  M2_STAMP(0, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
  const int gs = p.ctr_snap[0], ls = p.ctr_snap[1];   // taken by launch 1: nobody reads state[] while it is advanced
  const int B = p.B, C = p.C;
  const int sub = tid >> 5, l = tid & 31;
  const int i = 8 * tb + sub;
  const int qn = mlp2_next_slot(p.q);
  if (tb == 0 && tid == 0) {
    int* rec = p.wlog + (size_t)DTF_WLOG_INTS * (ls % p.log_cap);
    rec[0] = gs;       // value of global_step before this step's increment
    rec[1] = ls;
    if (MODE == 3) {
      // the ps-side counters: global_step next to shard 0, one "updates applied" counter per shard
      const DtfMlp2Async& as = p.async_;
      const int g2 = __hip_atomic_fetch_add(as.counters[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
      for (int s = 0; s < as.shards; ++s)
        __hip_atomic_fetch_add(as.counters[s] + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      t.state[0] = g2;   // global_step as the ps holds it after this worker's update
      if (as.astate) {
        // the bookkeeping of async_push_pull_kernel (p2p_allreduce.hip), for the push inside this launch
        const int ls1 = ls + 1;                          // steps of this worker including this one
        const bool pulled = mlp2_async_pulls(as, ls1);
        const int unseen = g2 - 1 - (as.astate[0] + as.astate[1]);
        if (as.alog && as.alog_cap > 0) {
          int* ar = as.alog + 2 * (size_t)(ls % as.alog_cap);
          ar[0] = g2;
          ar[1] = unseen < 0 ? 0 : unseen;
        }
        if (pulled) { as.astate[0] = g2; as.astate[1] = 0; }
        else as.astate[1] += 1;
        if (as.bound >= 0) {
          if (as.astate[6 + (ls1 & 1)] != 0 && !(as.pull_every <= 1 || (ls1 % as.pull_every) == 0)) as.astate[2] += 1;
          int force_next = (!pulled && unseen >= (as.bound > 0 ? as.bound : 1)) ? 1 : 0;
          int* clocks = as.counters[0] + 4;
          __hip_atomic_store(clocks + as.rank, ls1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (as.astate[5] == 0 && as.world > 1) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            bool waited = false;
            for (;;) {
              int slowest = 0x7fffffff;
              for (int j = 0; j < as.world; ++j)
                if (j != as.rank) slowest = min(slowest, __hip_atomic_load(clocks + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
              if (ls1 - slowest <= as.bound) break;
              waited = true;
              if (__builtin_amdgcn_s_memrealtime() - t0 > as.timeout_ticks) {
                as.astate[5] = 1;
                break;
              }
              __builtin_amdgcn_s_sleep(16);
            }
            if (waited) {
              as.astate[3] += 1;
              force_next = 1;                            // the others pushed while this worker waited
            }
          }
          as.astate[6 + ((ls1 + 1) & 1)] = force_next;   // read by the next launch; this launch reads the other slot
        }
      }
    } else {
      t.state[0] = gs + t.gs_inc;
    }
    t.state[1] = ls + 1;
  }
  if (i >= B) return;
  const long long pos64 = (long long)(ls + 1) * B + i;
  const long long epoch = pos64 / t.n_train;
  const int pos = (int)(pos64 - epoch * t.n_train);
  const int row = t.perm ? t.perm[(size_t)(epoch & 1) * t.n_train + pos] : pos;
  const f32x4* src = reinterpret_cast<const f32x4*>(p.X + (size_t)row * p.ldx);
  f32x4* dst = reinterpret_cast<f32x4*>(p.Xg + ((size_t)qn * MLP2_ROWS + i) * p.In);
  const int n4 = p.In >> 2;
  for (int base = 0; base < n4; base += 32 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = base + 32 * u + l;
      v[u] = src[k < n4 ? k : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = base + 32 * u + l;
      if (k < n4) dst[k] = v[u];
    }
  }
  if (l < 16) {
    const float v = p.labels[(size_t)row * C + (l < C ? l : 0)];
    p.labg[qn * MLP2_LABG_SLOT + l * MLP2_LABG_ROWS + i] = l < C ? v : 0.f;
  }
  if (l == 16) {
    int* rn = p.ridx + (size_t)qn * p.ridx_len;
    rn[i] = row;
    rn[B + i] = p.ridx[(size_t)p.q * p.ridx_len + i];   // the batch just trained on becomes the accuracy-only rows
  }
  M2_STAMP(6, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
}

// Softmax gradient of batch row `row` (one row per lane, classes in registers), written to LDS in the
// permuted class order the dz1 MFMAs read: a 16-byte read at 4*g returns classes g, 4+g, 8+g, 12+g.
template <int CP>
__device__ __forceinline__ void mlp2_dz2_row(const float (&z)[CP], const float (&lab)[CP], int C, bool row_ok, int loss_mode,
                                             float inv_b, float* dz_row) {
  const float NEG_INF = -__builtin_huge_valf();
  float zmax = NEG_INF;
#pragma unroll
  for (int c = 0; c < CP; ++c) zmax = fmaxf(zmax, c < C ? z[c] : NEG_INF);
  float e1[CP], s1 = 0.f;
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    e1[c] = c < C ? __expf(z[c] - zmax) : 0.f;
    s1 += e1[c];
  }
  const float inv1 = __builtin_amdgcn_rcpf(s1);
  float dz[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) dz[c] = 0.f;
  if (loss_mode == 0) {
    // softmax applied a second time on the probabilities (the reference feeds y as "logits")
    float y[CP], ymax = NEG_INF;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      y[c] = e1[c] * inv1;
      ymax = fmaxf(ymax, c < C ? y[c] : NEG_INF);
    }
    float e2[CP], s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      e2[c] = c < C ? __expf(y[c] - ymax) : 0.f;
      s2 += e2[c];
    }
    const float inv2 = __builtin_amdgcn_rcpf(s2);
    float dy[CP], dot = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      dy[c] = (e2[c] * inv2 - lab[c]) * inv_b;
      dot += dy[c] * y[c];
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) dz[c] = (dy[c] - dot) * y[c];
  } else {
#pragma unroll
    for (int c = 0; c < CP; ++c) dz[c] = (e1[c] * inv1 - lab[c]) * inv_b;
  }
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = q4 + 4 * j;
      v[j] = (row_ok && c < C) ? dz[c] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dz_row + 4 * q4) = v;
  }
}
__device__ __forceinline__ int dz_slot(int c) { return 4 * (c & 3) + (c >> 2); }

// One weight-gradient tile per wave.  IS2 = false: tiles of dW1 (inputs x hidden), dz1 rebuilt in registers;
// IS2 = true: tiles of dW2 (hidden x classes).  `tile0` is the first tile of this workgroup within its kind.
// MODE (compile time, = DtfMlp2Args.mode): the exchange / push code of modes 2 and 3 must not cost the plain
// modes registers or instructions (measured: 0.3 us per step when `mode` was a run-time branch).
template <int NJ, int NRB, bool IS2, int MODE>
__device__ __forceinline__ void mlp2_train_block(const DtfMlp2Args& p, int tile0, int Rp, unsigned zbytes, float* dzs,
                                                 int wgid, int* peer_timeout) {
  constexpr int CP = 4 * NJ;                        // classes padded to whole MFMA k-steps
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
  const unsigned h_bytes = (unsigned)(((size_t)(MLP2_ROWS - 1) * p.ldh + Hd) * 4u);   // H holds >= MLP2_ROWS rows
  M2_STAMP(0, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));

  // ---- operands of the softmax-gradient pass: lane tid < 16*NRB owns batch row tid ----
  const bool p1_wave = wave * 64 < 16 * NRB;        // wave-uniform
  float lab[CP], za[CP][4], zb[CP][4];
  if (p1_wave) {
    // no guards: slab entries of padding classes / rows are zero, and so are label rows beyond the batch
    const rsrc_t rz = make_rsrc(p.zpart, zbytes), rl = make_rsrc(p.labg, 3u * MLP2_LABG_SLOT * 4u);
    const unsigned zv = tid < Rp ? (unsigned)tid * 32u : kOOB, lv = (unsigned)tid * 4u;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      bld4_s(rz, zv, (unsigned)(c * Rp) * 32u, za[c]);
      bld4_s(rz, zv + 16u, (unsigned)(c * Rp) * 32u, zb[c]);
      lab[c] = bld_s(rl, lv, (unsigned)(p.q * MLP2_LABG_SLOT + c * MLP2_LABG_ROWS) * 4u);
    }
  }
  // ---- operands of the tile, requested before anything waits on the loads above ----
  float* const Wt = IS2 ? p.W2 : p.W1;
  float* const bt = IS2 ? p.b2 : p.b1;
  const rsrc_t rwt = make_rsrc(Wt, (unsigned)Mo * (unsigned)N * 4u), rbt = make_rsrc(bt, (unsigned)N * 4u);
  const rsrc_t rh = make_rsrc(p.H, h_bytes);
  float av[NRB][4], hm[NRB][4], wb[4] = {0.f, 0.f, 0.f, 0.f}, wold[4] = {0.f, 0.f, 0.f, 0.f}, bold = 0.f;
  {
    // Row 16*rb + 4*g + j of the A operand feeds MFMA j of row block rb.  Rows beyond the batch are zero
    // in Xg (padded slots) and harmless in H (their dz2 is zero), so only the column guard remains.
    const unsigned slot_bytes = (unsigned)MLP2_ROWS * (unsigned)p.In * 4u;
    const rsrc_t ra = IS2 ? rh : make_rsrc(p.Xg + (size_t)p.q * MLP2_ROWS * p.In, slot_bytes);
    const unsigned lda4 = (unsigned)(IS2 ? p.ldh : p.In) * 4u;
    const unsigned av_base = am_ok ? (unsigned)(4 * g) * lda4 + (unsigned)(m0 + fr) * 4u : kOOB;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int j = 0; j < 4; ++j) av[rb][j] = bld_s(ra, av_base, (unsigned)(16 * rb + j) * lda4);
  }
  if (!IS2) {
    const rsrc_t rw = make_rsrc(p.W2s, (unsigned)Hd * (unsigned)C * 4u);
    const unsigned ldh4 = (unsigned)p.ldh * 4u;
    const unsigned hm_base = bn_ok ? (unsigned)(4 * g) * ldh4 + (unsigned)(n0 + fr) * 4u : kOOB;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) hm[rb][r] = bld_s(rh, hm_base, (unsigned)(16 * rb + r) * ldh4);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      wb[j] = bld(rw, (bn_ok && 4 * j + g < C) ? ((unsigned)(n0 + fr) * (unsigned)C + (unsigned)(4 * j + g)) * 4u : kOOB);
  }
  // mode 3: where this lane's elements live in the masters (and, Adam, in the moments beside them).  The values are
  // requested further down, once the operand loads above have been consumed: requested here, the uncached reads sat in
  // front of them in the memory pipeline and every later stage of the workgroup started 2 us late.
  float* mptr[4] = {nullptr, nullptr, nullptr, nullptr};
  float* mbptr = nullptr;
  long long moff[4] = {0, 0, 0, 0}, mboff = 0;
  int msh[4] = {0, 0, 0, 0}, mbsh = 0;
  bool a_pull = true;
  if (MODE == 3) {
    const long long slice = p.async_.slice;
    const long long wbase = (long long)(Wt - p.async_.params_base), bbase = (long long)(bt - p.async_.params_base);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      const bool ok = row < Mo && bn_ok;
      const long long idx = ok ? wbase + (long long)row * N + n0 + fr : 0;
      msh[r] = p.async_.shards == 1 ? 0 : (int)(idx / slice);
      moff[r] = idx - (long long)msh[r] * slice;
      mptr[r] = p.async_.master[msh[r]] + moff[r];
    }
    const long long bidx = (mt == 0 && g == 0 && bn_ok) ? bbase + n0 + fr : 0;
    mbsh = p.async_.shards == 1 ? 0 : (int)(bidx / slice);
    mboff = bidx - (long long)mbsh * slice;
    mbptr = p.async_.master[mbsh] + mboff;
    a_pull = mlp2_async_pulls(p.async_, p.ctr_snap[1] + 1);
  }
  int seq = 0;
  bool dead = false;                                // the transport has failed earlier in this run: do not wait again
  if (MODE == 2 || MODE == 4) {
    seq = p.ctr_snap[1] + 1;                        // number of this step, the same on every rank
    dead = __hip_atomic_load(MODE == 2 ? p.peer.err : p.shard.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (tid == 0) *peer_timeout = 0;
  }
  // mode 4: only the rank that owns this workgroup's tiles needs their old values
  const bool owner4 = MODE == 4 && (wgid % p.shard.world) == p.shard.rank;      // workgroup-uniform
  if (MODE == 1 || MODE == 2 || MODE == 3 || owner4) {     // (mode 3: the replica's values, for a step that does not pull)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      wold[r] = bld(rwt, (row < Mo && bn_ok) ? ((unsigned)row * (unsigned)N + (unsigned)(n0 + fr)) * 4u : kOOB);
    }
    bold = bld(rbt, (mt == 0 && g == 0 && bn_ok) ? (unsigned)(n0 + fr) * 4u : kOOB);
  }
  M2_STAMP(1, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));

  // ---- softmax gradient of the whole batch into LDS (recomputed by every workgroup) ----
  if (p1_wave) {
    float z[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) z[c] = sum8(za[c], zb[c]);
    mlp2_dz2_row<CP>(z, lab, C, tid < B, p.loss_mode, p.inv_b, dzs + tid * 20);
  }
  // mode 3: the masters' values of this tile (uncached, possibly remote), requested here - behind every operand load in the
  // memory pipeline, in front of the barrier: the read runs beside the barrier and the products below
  float mw[4] = {0.f, 0.f, 0.f, 0.f}, mbw = 0.f, mm[4] = {0.f, 0.f, 0.f, 0.f}, mv[4] = {0.f, 0.f, 0.f, 0.f}, mbm = 0.f, mbv = 0.f;
  int mt_cnt[4] = {0, 0, 0, 0}, mbt_cnt = 0;
  if (MODE == 3 && has_tile) {
    const DtfMlp2Async& as = p.async_;
#pragma unroll
    for (int r = 0; r < 4; ++r) mw[r] = __builtin_nontemporal_load(mptr[r]);
    mbw = __builtin_nontemporal_load(mbptr);
    if (as.optimizer == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mm[r] = __builtin_nontemporal_load(as.m[msh[r]] + moff[r]);
        mv[r] = __builtin_nontemporal_load(as.v[msh[r]] + moff[r]);
        mt_cnt[r] = as.applied_snap[msh[r]];
      }
      mbm = __builtin_nontemporal_load(as.m[mbsh] + mboff);
      mbv = __builtin_nontemporal_load(as.v[mbsh] + mboff);
      mbt_cnt = as.applied_snap[mbsh];
    }
  }

  M2_STAMP(3, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
  __syncthreads();
  M2_STAMP(4, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
  // (mode 2: waves without a tile go on to the barriers of the exchange; their loads are out of range, i.e. zero,
  // and every store below is guarded)
  if (!has_tile && MODE != 2 && MODE != 4) return;
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
  M2_STAMP(5, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
  float* const dWt = IS2 ? p.dW2 : p.dW1;
  float* const dbt = IS2 ? p.db2 : p.db1;
  float gv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) gv[r] = acc0[r] + acc1[r];
  if (mt == 0) {  // wave-uniform: this tile also owns the bias gradient of its 16 columns
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
  }
  const bool b_lane = mt == 0 && g == 0 && bn_ok;
  if (MODE == 4) {
    // ---- sharded exchange: tile -> owner (reduce-scatter), updated tile -> everybody (all-gather) ----
    const DtfMlp2Shard& sh = p.shard;
    const int me = sh.rank, world = sh.world;
    const int owner = wgid % world;
    size_t woff[4];
    bool wok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      wok[r] = row < Mo && bn_ok;
      woff[r] = wok[r] ? (size_t)(Wt - sh.params_base) + (size_t)row * N + n0 + fr : 0;
    }
    const size_t boff = b_lane ? (size_t)(bt - sh.params_base) + n0 + fr : 0;
    // Adam: the owner's moments of these tiles are requested before the wait for the peers
    float am[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f}, abm = 0.f, abv = 0.f;
    if (owner4 && sh.adam) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        am[r] = sh.m[woff[r]];
        av[r] = sh.v[woff[r]];
      }
      abm = sh.m[boff];
      abv = sh.v[boff];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (!owner4) {
      // my contribution into the owner's inbox (row `me`), then the flag: stores and flag are system scope on
      // uncached memory, the flag leaves after every store of the workgroup has been acknowledged
      float* in = reinterpret_cast<float*>(sh.region[owner] + sh.inbox_off) + (size_t)me * sh.n;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (wok[r]) __hip_atomic_store(in + woff[r], gv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (b_lane) __hip_atomic_store(in + boff, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(reinterpret_cast<int*>(sh.region[owner] + sh.inflag_off) + me * DTF_MLP2_MAX_WG + wgid, seq,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the owner's answer: this workgroup's tiles, updated, in MY outbox
        int* f = reinterpret_cast<int*>(sh.region[me] + sh.outflag_off) + wgid;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
          if (dead || __builtin_amdgcn_s_memrealtime() - t0 > sh.timeout_ticks) {
            *peer_timeout = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      }
      __syncthreads();
      if (*peer_timeout) {                          // parameters stay untouched; the host sees the flag
        if (tid == 0) *sh.err = 1;
        return;
      }
      const float* ob = reinterpret_cast<const float*>(sh.region[me] + sh.outbox_off);
      float nv[4], nb = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) nv[r] = __hip_atomic_load(ob + woff[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (b_lane) nb = __hip_atomic_load(ob + boff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (wok[r]) Wt[(size_t)(m0 + 4 * g + r) * N + n0 + fr] = nv[r];
      if (b_lane) bt[n0 + fr] = nb;
      M2_STAMP(6, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
      return;
    }
    // owner: wait for this workgroup's tiles of every peer (they sit in local memory), add in rank order
    if (tid < world && tid != me) {
      int* f = reinterpret_cast<int*>(sh.region[me] + sh.inflag_off) + tid * DTF_MLP2_MAX_WG + wgid;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
        if (dead || __builtin_amdgcn_s_memrealtime() - t0 > sh.timeout_ticks) {
          *peer_timeout = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (*peer_timeout) {
      if (tid == 0) *sh.err = 1;
      return;
    }
    const float* inb = reinterpret_cast<const float*>(sh.region[me] + sh.inbox_off);
    float pv[8][4], pb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* src = inb + (size_t)(k < world ? k : me) * sh.n;      // rows beyond the group: read again, not added
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[k][r] = __hip_atomic_load(src + woff[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      pb[k] = __hip_atomic_load(src + boff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float sv[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < world) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] += k == me ? gv[r] : pv[k][r];
        sb += k == me ? bsum : pb[k];
      }
    float nv[4], nb;
    if (sh.adam) {
      // the owner keeps m and v of its tiles (the slot variables live where the variable lives, as on a ps task);
      // same element update as the flat optimizer kernels
      const float t = (float)(seq + sh.adam_t_offset);
      const float bc1 = 1.f - powf(sh.b1, t), bc2 = 1.f - powf(sh.b2, t);
      const float sq = sqrtf(bc2);
      const float lr_t = sh.adam_lr * sq / bc1, eps_t = sh.eps * sq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        nv[r] = wold[r];
        dtf_adam_one(nv[r], sv[r] * sh.grad_scale, am[r], av[r], lr_t, sh.b1, sh.b2, eps_t);
        if (wok[r]) {
          sh.m[woff[r]] = am[r];
          sh.v[woff[r]] = av[r];
        }
      }
      nb = bold;
      dtf_adam_one(nb, sb * sh.grad_scale, abm, abv, lr_t, sh.b1, sh.b2, eps_t);
      if (b_lane) {
        sh.m[boff] = abm;
        sh.v[boff] = abv;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) nv[r] = wold[r] - p.lr * sv[r];
      nb = bold - p.lr * sb;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < world && k != me) {
        float* ob = reinterpret_cast<float*>(sh.region[k] + sh.outbox_off);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (wok[r]) __hip_atomic_store(ob + woff[r], nv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (b_lane) __hip_atomic_store(ob + boff, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (wok[r]) Wt[(size_t)(m0 + 4 * g + r) * N + n0 + fr] = nv[r];
    if (b_lane) bt[n0 + fr] = nb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < world && tid != me)
      __hip_atomic_store(reinterpret_cast<int*>(sh.region[tid] + sh.outflag_off) + wgid, seq, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    M2_STAMP(6, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
    return;
  }
  if (MODE == 2) {
    // ---- exchange with the same workgroup of every peer, then the averaged update ----
    const int me = p.peer.rank, world = p.peer.world;
    size_t woff[4];
    bool wok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      wok[r] = row < Mo && bn_ok;
      woff[r] = wok[r] ? (size_t)(dWt - p.peer.slot[me]) + (size_t)row * N + n0 + fr : 0;
      // system-scope stores (write-through past every cache and write buffer of this XCD): their acknowledgement,
      // awaited below, then means "in memory", which is what a peer - another GPU, or another XCD - reads
      if (wok[r]) __hip_atomic_store(&dWt[(size_t)row * N + n0 + fr], gv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const size_t boff = b_lane ? (size_t)(dbt - p.peer.slot[me]) + n0 + fr : 0;
    if (b_lane) __hip_atomic_store(&dbt[n0 + fr], bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // The slots and flags are uncached memory: a store is in memory (not in a cache) once it is acknowledged, so
    // "acknowledged, then barrier, then flag" publishes the tile.  No system-scope fence: it would write back and
    // invalidate the whole L2 once per workgroup, under the other workgroups' feet.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < world && tid != me) {
      __hip_atomic_store(p.peer.wgflags[tid] + ((p.q * 8 + me) * DTF_MLP2_MAX_WG + wgid), seq, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
      int* f = p.peer.wgflags[me] + ((p.q * 8 + tid) * DTF_MLP2_MAX_WG + wgid);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      // step numbers only grow; a wall-clock limit instead of an endless spin
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq < 0) {
        if (dead || __builtin_amdgcn_s_memrealtime() - t0 > p.peer.timeout_ticks) {
          *peer_timeout = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      // acquire once per workgroup, after the last flag: invalidates (writes nothing back) whatever this CU or
      // its L2 may hold of the peers' slots, however the peer mapping is cached on the machine at hand
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    if (*peer_timeout) {                            // parameters stay untouched; the host sees the flag
      if (tid == 0) *p.peer.err = 1;
      return;
    }
    // every rank adds the N tiles in rank order (its own from registers): bit-identical replicas.  Ranks beyond
    // the group read this rank's slot again, so the loads need no branch; they are not added.
    float pv[8][4], pb[8];
    if (world > 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float* src = p.peer.slot[k < world ? k : me];
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[k][r] = __hip_atomic_load(src + woff[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        pb[k] = __hip_atomic_load(src + boff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    float sv[4] = {0.f, 0.f, 0.f, 0.f}, sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < world) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] += k == me ? gv[r] : pv[k][r];
        sb += k == me ? bsum : pb[k];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (wok[r]) Wt[(size_t)(m0 + 4 * g + r) * N + n0 + fr] = wold[r] - p.lr * sv[r];
    if (b_lane) bt[n0 + fr] = bold - p.lr * sb;
    M2_STAMP(6, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * g + r;
    if (row < Mo && bn_ok) {
      const size_t off = (size_t)row * N + n0 + fr;
      if (MODE == 3) {
        // push: lock-free read-modify-write of the master, wherever it lives (and of the moments beside it)
        float w;
        if (p.async_.optimizer == 1) {
          w = mlp2_async_adam(p.async_, p.lr, mw[r], gv[r], mm[r], mv[r], mt_cnt[r]);
          __builtin_nontemporal_store(mm[r], p.async_.m[msh[r]] + moff[r]);
          __builtin_nontemporal_store(mv[r], p.async_.v[msh[r]] + moff[r]);
        } else {
          w = mw[r] - p.lr * gv[r];
        }
        __builtin_nontemporal_store(w, mptr[r]);
        // pull: the master's value is this worker's replica; between pulls the replica takes this worker's own update
        Wt[off] = a_pull ? w : wold[r] + (w - mw[r]);
      } else if (MODE == 1) {
        Wt[off] = wold[r] - p.lr * gv[r];
      } else {
        dWt[off] = gv[r];
      }
    }
  }
  if (b_lane) {
    if (MODE == 3) {
      float w;
      if (p.async_.optimizer == 1) {
        w = mlp2_async_adam(p.async_, p.lr, mbw, bsum, mbm, mbv, mbt_cnt);
        __builtin_nontemporal_store(mbm, p.async_.m[mbsh] + mboff);
        __builtin_nontemporal_store(mbv, p.async_.v[mbsh] + mboff);
      } else {
        w = mbw - p.lr * bsum;
      }
      __builtin_nontemporal_store(w, mbptr);
      bt[n0 + fr] = a_pull ? w : bold + (w - mbw);
    } else if (MODE == 1) {
      bt[n0 + fr] = bold - p.lr * bsum;
    } else {
      dbt[n0 + fr] = bsum;
    }
  }
  M2_STAMP(6, dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x));
}

// workgroups of launch 2 (role index): nb1 blocks of dW1 tiles | nb2 blocks of dW2 tiles | R16 stat | ceil(B/8) tail
template <int NJ, int NRB, int MODE>
__global__ __launch_bounds__(256) void mlp2_bwd_kernel(DtfMlp2Args p) {
  __shared__ float dzs[128 * 20];
  __shared__ int peer_timeout;
  const int R16 = (p.R + 15) >> 4, Rp = R16 << 4;
  const int NT = (p.Hd + 15) >> 4;
  const int nb1 = (((p.In + 15) >> 4) * NT + 3) >> 2, nb2 = (NT + 3) >> 2;
  const unsigned zbytes = 16u * (unsigned)Rp * 8u * 4u;
  // Workgroup b of a launch runs on XCD b mod 8 and every XCD has its own L2; what a launch finds there is what that
  // XCD's workgroups of the launch before it touched.  The role index is XCD-major (blocks 0, 8, 16, ... get indices
  // 0, 1, 2, ...) in both launches of the step - neighbouring weight-gradient tiles (same 16 input features, adjacent
  // hidden tiles) and neighbouring forward tiles (same batch rows) share an L2 - with the tile workgroups first.
  // Measured (2000 steps, one box, us/step; one worker / sharded / replicated exchange / async on one rank):
  // plain block order 11.75 / 13.17 / 12.76 / 14.00; XCD-major, stat | dW1 | dW2 | tail 10.92 / 12.53 / 11.81 /
  // 13.17; dW1 | dW2 | stat | tail (kept) 10.93 / 12.23 / 11.79 / 13.23; tail | stat | dW1 | dW2 11.09 / 12.70 /
  // 12.22 / 13.37; forward tiles column-major instead of row-block-major 11.04 / 12.67 / 11.97 / 13.27.
  const int b = dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x);
  if (b < nb1) mlp2_train_block<NJ, NRB, false, MODE>(p, b * 4, Rp, zbytes, dzs, b, &peer_timeout);
  else if (b < nb1 + nb2) mlp2_train_block<NJ, NRB, true, MODE>(p, (b - nb1) * 4, Rp, zbytes, dzs, b, &peer_timeout);
  else if (b < nb1 + nb2 + R16) mlp2_stat_block(p, b - nb1 - nb2, Rp, zbytes);
  else mlp2_tail_block<MODE>(p, b - R16 - nb1 - nb2);
}

static int mlp2_validate(const DtfMlp2Args* a) {
  if (!a || !a->X || !a->labels || !a->Xg || !a->labg || !a->ridx || !a->H || !a->zpart || !a->W2s || !a->ctr_snap || !a->wlog)
    return DTF_ERR_ARG;
  if (!a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->tail.state) return DTF_ERR_ARG;
  if (a->mode < 0 || a->mode > 4) return DTF_ERR_ARG;
  if (a->mode == 4) {
    const DtfMlp2Shard& sh = a->shard;
    if (sh.world < 1 || sh.world > 8 || sh.rank < 0 || sh.rank >= sh.world || !sh.err || !sh.params_base || sh.n <= 0)
      return DTF_ERR_ARG;
    for (int k = 0; k < sh.world; ++k)
      if (!sh.region[k]) return DTF_ERR_ARG;
    if (sh.adam && (!sh.m || !sh.v || !(sh.adam_lr > 0.f) || !(sh.grad_scale > 0.f) || !(sh.b1 >= 0.f && sh.b1 < 1.f) ||
                    !(sh.b2 >= 0.f && sh.b2 < 1.f) || !(sh.eps > 0.f)))
      return DTF_ERR_ARG;
    // the four parameter tensors must be views into the flat buffer whose offsets address the boxes
    const float* lo = sh.params_base;
    const float* hi = lo + sh.n;
    if (a->W1 < lo || a->b1 < lo || a->W2 < lo || a->b2 < lo) return DTF_ERR_ARG;
    if (a->W1 + (size_t)a->In * a->Hd > hi || a->b1 + a->Hd > hi || a->W2 + (size_t)a->Hd * a->C > hi || a->b2 + a->C > hi)
      return DTF_ERR_SHAPE;
    const int NTv = dtf_cdiv(a->Hd, 16);
    if (dtf_cdiv(dtf_cdiv(a->In, 16) * NTv, 4) + dtf_cdiv(NTv, 4) > DTF_MLP2_MAX_WG) return DTF_ERR_SHAPE;
  }
  if ((a->mode == 0 || a->mode == 2) && (!a->dW1 || !a->db1 || !a->dW2 || !a->db2)) return DTF_ERR_ARG;
  if (a->mode == 3) {
    const DtfMlp2Async& as = a->async_;
    if (as.shards < 1 || as.shards > 8 || as.slice <= 0 || !as.params_base) return DTF_ERR_ARG;
    if (as.optimizer != 0 && as.optimizer != 1) return DTF_ERR_ARG;
    for (int s = 0; s < as.shards; ++s) {
      if (!as.master[s] || !as.counters[s]) return DTF_ERR_ARG;
      if (as.optimizer == 1 && (!as.m[s] || !as.v[s])) return DTF_ERR_ARG;
    }
    if (as.optimizer == 1 && !as.applied_snap) return DTF_ERR_ARG;
    if (as.optimizer == 1 && (!(as.b1 >= 0.f && as.b1 < 1.f) || !(as.b2 >= 0.f && as.b2 < 1.f) || !(as.eps > 0.f))) return DTF_ERR_ARG;
    if (as.pull_every < 1 || (as.alog && !as.astate) || (as.alog && as.alog_cap <= 0)) return DTF_ERR_ARG;
    if (as.bound >= 0 && (!as.astate || as.world < 1 || as.world > 8 || as.rank < 0 || as.rank >= as.world)) return DTF_ERR_ARG;
    // the four parameter tensors must be views into the flat buffer the slices partition
    const float* lo = as.params_base;
    const float* hi = lo + as.slice * as.shards;
    if (a->W1 < lo || a->b1 < lo || a->W2 < lo || a->b2 < lo) return DTF_ERR_ARG;
    if (a->W1 + (size_t)a->In * a->Hd > hi || a->b1 + a->Hd > hi || a->W2 + (size_t)a->Hd * a->C > hi || a->b2 + a->C > hi) return DTF_ERR_SHAPE;
  }
  if (a->mode == 2) {
    const DtfMlp2Peer& pe = a->peer;
    if (pe.world < 1 || pe.world > 8 || pe.rank < 0 || pe.rank >= pe.world || !pe.err) return DTF_ERR_ARG;
    for (int k = 0; k < pe.world; ++k)
      if (!pe.slot[k] || !pe.wgflags[k]) return DTF_ERR_ARG;
    // the four gradient tensors must lie in this rank's slot (their offsets address the peers' slots too)
    const float* base = pe.slot[pe.rank];
    if (a->dW1 < base || a->db1 < base || a->dW2 < base || a->db2 < base) return DTF_ERR_ARG;
    const int NTv = dtf_cdiv(a->Hd, 16);
    if (dtf_cdiv(dtf_cdiv(a->In, 16) * NTv, 4) + dtf_cdiv(NTv, 4) > DTF_MLP2_MAX_WG) return DTF_ERR_SHAPE;
  }
  if (a->In <= 0 || a->Hd <= 0 || a->Hd > 128 || a->C <= 0 || a->C > 16 || a->B <= 0 || a->B > MLP2_LABG_ROWS) return DTF_ERR_SHAPE;
  if (a->R != a->B && a->R != 2 * a->B) return DTF_ERR_SHAPE;
  if (a->q < 0 || a->q > 2) return DTF_ERR_ARG;
  if (a->ldx < a->In || a->ldh < a->Hd || a->log_cap <= 0) return DTF_ERR_SHAPE;
  if ((a->In & 3) || (a->ldx & 3)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->X) || !dtf_aligned16(a->Xg) || !dtf_aligned16(a->zpart)) return DTF_ERR_ALIGN;
  if (a->ridx_len < 2 * a->B) return DTF_ERR_SHAPE;
  if ((uint64_t)3 * MLP2_ROWS * a->In * 4u >= 0xFFFFFFF0u || a->h_rows < MLP2_ROWS || a->h_rows < a->R) return DTF_ERR_SHAPE;
  if (a->tail.n_train <= 0 || !a->tail.enable) return DTF_ERR_SHAPE;
  return DTF_OK;
}

DTF_EXPORT int dtf_mlp2_fwd(const DtfMlp2Args* a, hipStream_t s) {
  const int rc = mlp2_validate(a);
  if (rc != DTF_OK) return rc;
  hipLaunchKernelGGL(mlp2_fwd_kernel, dim3(dtf_cdiv(a->Hd, 16), dtf_cdiv(a->R, 16)), dim3(1024), 0, s, *a);
  return dtf_check_launch();
}

template <int NJ, int MODE>
static void mlp2_launch_bwd_mode(const DtfMlp2Args& a, int nblk, hipStream_t s) {
  const int nrb = dtf_cdiv(a.B, 16);
  if (nrb <= 2) hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 2, MODE>), dim3(nblk), dim3(256), 0, s, a);
  else if (nrb <= 4) hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 4, MODE>), dim3(nblk), dim3(256), 0, s, a);
  else if (nrb <= 7) hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 7, MODE>), dim3(nblk), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((mlp2_bwd_kernel<NJ, 8, MODE>), dim3(nblk), dim3(256), 0, s, a);
}

template <int NJ>
static void mlp2_launch_bwd(const DtfMlp2Args& a, int nblk, hipStream_t s) {
  switch (a.mode) {
    case 0: mlp2_launch_bwd_mode<NJ, 0>(a, nblk, s); break;
    case 1: mlp2_launch_bwd_mode<NJ, 1>(a, nblk, s); break;
    case 2: mlp2_launch_bwd_mode<NJ, 2>(a, nblk, s); break;
    case 3: mlp2_launch_bwd_mode<NJ, 3>(a, nblk, s); break;
    default: mlp2_launch_bwd_mode<NJ, 4>(a, nblk, s); break;
  }
}

DTF_EXPORT int dtf_mlp2_bwd(const DtfMlp2Args* a, hipStream_t s) {
  const int rc = mlp2_validate(a);
  if (rc != DTF_OK) return rc;
  const int NT = dtf_cdiv(a->Hd, 16);
  const int nblk = dtf_cdiv(a->R, 16) + dtf_cdiv(dtf_cdiv(a->In, 16) * NT, 4) + dtf_cdiv(NT, 4) + dtf_cdiv(a->B, 8);
  if (a->C <= 12) mlp2_launch_bwd<3>(*a, nblk, s);
  else mlp2_launch_bwd<4>(*a, nblk, s);
  return dtf_check_launch();
}
