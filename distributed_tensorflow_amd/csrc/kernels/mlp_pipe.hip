// One-launch training step of the reference network x -> relu(x W1 + b1) -> softmax(h W2 + b2)
// (math: /root/reference/distributed.py:78-89 and its autodiff), exact fp32 MFMA throughout, for ONE worker
// with plain SGD.
//
// The two-launch step (mlp2_fused.hip) is bound by two dependent device-wide phases per step: the forward
// product needs every tile of the updated W1, and the weight gradient needs the whole hidden layer (through the
// logits).  Each phase change costs a kernel boundary (~2 us) plus a cold first touch of what the other phase
// wrote.  This file removes one of the two by never waiting for W1 to be MATERIALISED.  With X'(s) the batch of
// step s extended by a column of ones (so that b1 is row `In` of W1' and one product covers both),
//
//     W1'(s) = W1'(s-2) - lr X'(s-2)^T dz1(s-2) - lr X'(s-1)^T dz1(s-1)
//     z1(s)  = X'(s) W1'(s)
//            = [ X'(s) W1'(s-2) - lr G(s,s-2) dz1(s-2) ]  -  lr G(s,s-1) dz1(s-1),      G(a,b) = X'(a) X'(b)^T,
//
// the bracket (called Q(s)) depends on nothing newer than step s-2 and is computed, spread over the chip, one
// launch EARLIER; the Gram matrices G depend on the data only and are computed one launch earlier still.  What is
// left on the dependent path of step s is small: per 16 batch rows one workgroup computes
// z1 = Q - lr G dz1(s-1) (a [16 x B] x [B x H] product), the logits, the loss, dz2 and dz1(s) - all row-local - and
// every other quantity of the step is produced by workgroups of the same launch that nobody in that launch waits
// for.  One launch per step, one kernel boundary per step.  The weight matrix itself follows one launch behind
// (it is only read by the Q products), the output layer (101 x 10) is brought up to date by every row workgroup
// from per-row-block partial gradients of the previous launch, and a run of steps ends with a flush launch that
// writes W1, b1, W2, b2 of the last step into the flat parameter buffer; it begins with a prime launch that builds
// Q of its first step from that buffer.  The accuracy-only rows (previous batch with the updated parameters,
// distributed.py:148-149) need no product of their own: X'(s-1) W1'(s) = z1(s-1) - lr G(s-1,s-1) dz1(s-1), and
// z1(s-1) is what the row workgroups of the previous launch computed (they keep it in Z1).
//
// Launch of step s, slot q = s mod 8, generation a = s & 1 (p = 1 - a):
//   role L  row workgroups (train row blocks, then accuracy-only row blocks): the dependent path above; reads
//           Q[a] (accuracy-only rows: Z1[p]), GG[p], D1[p], W2m[p], P2[p]; writes D1[a] (= dz1(s)), Z1[a] (= z1(s)),
//           P2[a][row block], W2m[a], the log record
//   role Q  Q[p] (for step s+1) = X'(s+1) Wm[p] - lr G(s+1,s-1) D1[p]
//           (prime launch only: also Z1 of the batch before the first step from the flat parameters)
//   role G  GG[a] = { G(s,s), G(s+1,s), G(s+2,s) }, one 16x16 tile per workgroup
//   role W  Wm[a] = Wm[p] - lr X'(s-1)^T D1[p]            (flush: into W1 / b1 of the flat buffer)
//   role T  counters, log header; gather of the batch of step s+3 (rows, label rows, label indices) through the
//           row indices the previous launch selected, and selection of the row indices of step s+4 (no load of
//           this role depends on more than one other load)
//   role F  (flush only) W2, b2 of the flat buffer = W2m[p] - lr sum of P2[p]
// Everything a launch reads of generation p was written by earlier launches; everything it writes is generation a
// (or a slot no role of this launch reads), so the roles of one launch never depend on each other.
#include "mlp_pipe_common.h"

// Diagnostic build only (-DDTF_STAMPS): per-workgroup timestamps of the 100 MHz constant clock.
#ifdef DTF_STAMPS
__device__ unsigned long long* pipe_stamp_buf = nullptr;
#define PIPE_STAMP(slot)                                                                                                       \
  do {                                                                                                                         \
    if (threadIdx.x == 0 && pipe_stamp_buf) pipe_stamp_buf[(size_t)(gridDim.x <= 512 ? dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int dtf_set_stamp_buffer_pipe(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(pipe_stamp_buf), &p, sizeof(p));
}
// shader-clock counter into the same buffer (clock frequency under this load = cycles / elapsed constant-clock time)
#define PIPE_STAMP_CYCLES(slot)                                                                                      \
  do {                                                                                                               \
    if (threadIdx.x == 0 && pipe_stamp_buf) pipe_stamp_buf[(size_t)(gridDim.x <= 512 ? dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 8 + (slot)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PIPE_STAMP(slot) do { } while (0)
#define PIPE_STAMP_CYCLES(slot) do { } while (0)
#endif

// Two geometries of the operand-layout matrices.  TD = 8: matrices of 8 x 8 tiles (batch <= 128, hidden + 1 <= 128) -
// the reference shape, every constant a power of two the compiler folds.  TD = 16: 16 x 16 tiles (batch <= 256,
// hidden <= 256).  With hidden = 256 the extended hidden layer h' has a 17th tile that holds nothing but the constant
// column carrying b2: dz1, Q and Z1 have no such tile (their columns stop at Hd), only the output layer's rows do.
template <int TD>
struct PipeGeo {
  static constexpr int ROWS = 16 * TD;              // rows of a gathered-batch slot
  static constexpr int MAT = TD * TD * 256;         // floats per operand-layout matrix
  static constexpr int LABG = 16 * ROWS;            // floats per gathered-label slot
  static constexpr int NCH = TD == 16 ? 5 : 4;      // split-K chunks per wave in role Q
};
template <int TD>
__host__ __device__ __forceinline__ unsigned pfrag(int t0, int t1, int lane) { return (unsigned)(((t0 * TD + t1) * 64 + lane) * 4); }
// hidden tiles that Q / Z1 / D1 have
template <int TD>
__host__ __device__ __forceinline__ int pipe_ntq(const DtfPipeArgs& p) { return TD == 16 ? (p.Hd + 15) >> 4 : p.HP >> 4; }

struct PipeDims {
  int NT, KT, nL, nQ, nG, nW, nT, nF, nC;
};
template <int TD>
__host__ __device__ __forceinline__ long long pipe_copy_elems(const DtfPipeArgs& p) {
  return 2ll * ((long long)p.KAp * p.HP + (long long)p.HP * 16 + (long long)PipeGeo<TD>::MAT + (long long)TD * p.HP * 16);
}
template <int TD>
__host__ __device__ __forceinline__ PipeDims pipe_dims(const DtfPipeArgs& p, int nrb) {
  PipeDims d;
  const int post = p.R > p.B ? 1 : 0;
  d.NT = p.HP >> 4;
  d.KT = p.KAp >> 4;
  d.nL = (p.roles & PIPE_ROLE_L) ? nrb * (1 + post) : 0;
  d.nQ = (p.roles & PIPE_ROLE_Q) ? nrb * (1 + (p.prime ? post : 0)) * pipe_ntq<TD>(p) : 0;
  // G(s,s) is symmetric: only its tiles on and above the diagonal are computed (and stored in both orientations)
  d.nG = (p.roles & PIPE_ROLE_G) ? 2 * nrb * nrb + post * (nrb * (nrb + 1) / 2) : 0;
  d.nW = (p.roles & PIPE_ROLE_W) ? (d.KT * d.NT + 15) >> 4 : 0;
  d.nT = (p.roles & PIPE_ROLE_T) ? (p.B + 7) >> 3 : 0;
  d.nF = (p.roles & PIPE_ROLE_F) ? 1 : 0;
  d.nC = (p.roles & PIPE_ROLE_C) ? (int)((pipe_copy_elems<TD>(p) + 4095) / 4096) : 0;
  return d;
}

// ---------------------------------------------------------------------------------------------------------
// role Q: one 16x16 tile of Q(s+1) per 16-wave workgroup, split-K over the waves
// ---------------------------------------------------------------------------------------------------------
template <int NRB, int TD>
__device__ __forceinline__ void pipe_role_q(const DtfPipeArgs& p, int idx, float* smem) {
  using G = PipeGeo<TD>;
  constexpr int NCH = G::NCH;
  float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int NT = p.HP >> 4;
  const int NTQ = pipe_ntq<TD>(p);
  const int rbq = idx / NTQ, nt = idx - rbq * NTQ;
  const bool post = rbq >= NRB;                        // prime launch only: z1 of the batch before the first step
  const int rb = post ? rbq - NRB : rbq;
  const int m0 = rb * 16, n0 = nt * 16;
  const int a = p.q & 1, pg = a ^ 1;
  const int slot = post ? p.q : ((p.q + 1) & (PIPE_SLOTS - 1));
  const unsigned slot_bytes = (unsigned)G::ROWS * (unsigned)p.ldxa * 4u;
  const rsrc_t rx = make_rsrc(p.Xa + (size_t)slot * G::ROWS * p.ldxa, slot_bytes);
  const rsrc_t rw = make_rsrc(p.Wm + (size_t)pg * p.KAp * p.HP, (unsigned)p.KAp * (unsigned)p.HP * 4u);
  const rsrc_t rg = make_rsrc(p.GG + ((size_t)pg * 3 + 2) * G::MAT, G::MAT * 4u);
  const rsrc_t rd = make_rsrc(p.D1 + (size_t)pg * G::MAT, G::MAT * 4u);
  const int nX = p.KAp >> 4;
  float av[NCH][4], bv[NCH][4];
  float scale[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = wave + 16 * i;                       // wave-uniform
    scale[i] = 1.f;
    if (c < nX) {
      const unsigned k = (unsigned)(16 * c + 4 * g);
      bld4(rx, (unsigned)(m0 + fr) * (unsigned)p.ldxa * 4u + k * 4u, av[i]);
      if (p.prime) {
        // the working copy is being written by role C of this launch: read W1 / b1 of the flat buffer
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = (int)k + j, cc = n0 + fr;
          float v = 0.f;
          if (cc < p.Hd) {
            if (kk < p.In) v = p.W1[(size_t)kk * p.Hd + cc];
            else if (kk == p.In) v = p.b1[cc];
          }
          bv[i][j] = v;
        }
      } else {
        bld4(rw, pipe_wfrag(c, nt, NT, lane) * 4u, bv[i]);
      }
    } else if (c - nX < NRB && !p.prime) {       // (prime: no pending gradient)
      scale[i] = -p.lr;
      bld4(rg, pfrag<TD>(rb, c - nX, lane) * 4u, av[i]);
      bld4(rd, pfrag<TD>(nt, c - nX, lane) * 4u, bv[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) av[i][j] = bv[i][j] = 0.f;
    }
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    acc0 = MFMA16(av[i][0] * scale[i], bv[i][0], acc0);
    acc1 = MFMA16(av[i][1] * scale[i], bv[i][1], acc1);
    acc0 = MFMA16(av[i][2] * scale[i], bv[i][2], acc0);
    acc1 = MFMA16(av[i][3] * scale[i], bv[i][3], acc1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  PIPE_STAMP(1);
  __syncthreads();
  PIPE_STAMP(2);
  if (threadIdx.x < 256) {
    const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
    const int e_rl = 4 * (e_ln >> 4) + e_r, e_cl = e_ln & 15;
    const float s = pipe_reduce16(red, e_r, e_ln);
    float* out = post ? p.Z1 + (size_t)a * G::MAT : p.Q + (size_t)pg * G::MAT;
    out[pfrag<TD>(rb, nt, (e_rl >> 2) * 16 + e_cl) + (e_rl & 3)] = s;
  }
  PIPE_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------
// role G: one 16x16 tile of a Gram matrix X'(s+kind) X'(s)^T per 16-wave workgroup
// ---------------------------------------------------------------------------------------------------------
template <int NRB, int TD>
__device__ __forceinline__ void pipe_role_g(const DtfPipeArgs& p, int idx, float* smem) {
  using G = PipeGeo<TD>;
  float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  // kinds 1 and 2 first (NRB x NRB tiles each), then - with accuracy-only rows - the upper triangle of kind 0
  int kind, rbi, cbj;
  if (idx < 2 * NRB * NRB) {
    kind = 1 + idx / (NRB * NRB);
    const int rem = idx - (kind - 1) * NRB * NRB;
    rbi = rem / NRB;
    cbj = rem - rbi * NRB;
  } else {
    kind = 0;
    int t = idx - 2 * NRB * NRB;
    rbi = 0;
    while (t >= NRB - rbi) {
      t -= NRB - rbi;
      ++rbi;
    }
    cbj = rbi + t;
  }
  const int m0 = rbi * 16, n0 = cbj * 16;
  const int a = p.q & 1;
  const unsigned slot_bytes = (unsigned)G::ROWS * (unsigned)p.ldxa * 4u;
  const rsrc_t ra = make_rsrc(p.Xa + (size_t)((p.q + kind) & (PIPE_SLOTS - 1)) * G::ROWS * p.ldxa, slot_bytes);
  const rsrc_t rb = make_rsrc(p.Xa + (size_t)p.q * G::ROWS * p.ldxa, slot_bytes);
  const int nX = p.KAp >> 4;
  float av[4][4], bv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = wave + 16 * i;
    const unsigned k4 = (unsigned)(16 * c + 4 * g) * 4u;
    bld4(ra, c < nX ? (unsigned)(m0 + fr) * (unsigned)p.ldxa * 4u + k4 : kOOB, av[i]);
    bld4(rb, c < nX ? (unsigned)(n0 + fr) * (unsigned)p.ldxa * 4u + k4 : kOOB, bv[i]);
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc0 = MFMA16(av[i][0], bv[i][0], acc0);
    acc1 = MFMA16(av[i][1], bv[i][1], acc1);
    acc0 = MFMA16(av[i][2], bv[i][2], acc0);
    acc1 = MFMA16(av[i][3], bv[i][3], acc1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  PIPE_STAMP(1);
  __syncthreads();
  PIPE_STAMP(2);
  if (threadIdx.x < 256) {
    const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
    const int e_rl = 4 * (e_ln >> 4) + e_r, e_cl = e_ln & 15;
    const float s = pipe_reduce16(red, e_r, e_ln);
    float* out = p.GG + ((size_t)a * 3 + kind) * G::MAT;
    out[pfrag<TD>(rbi, cbj, (e_cl >> 2) * 16 + e_rl) + (e_cl & 3)] = s;
    if (kind == 0 && cbj != rbi) out[pfrag<TD>(cbj, rbi, (e_rl >> 2) * 16 + e_cl) + (e_rl & 3)] = s;    // the mirrored tile
  }
  PIPE_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------
// role W: Wm[a] = Wm[p] - lr X'(s-1)^T D1[p], one 16x16 tile per wave (16 tiles per workgroup)
// ---------------------------------------------------------------------------------------------------------
template <int NRB, int TD>
__device__ __forceinline__ void pipe_role_w(const DtfPipeArgs& p, int idx) {
  using G = PipeGeo<TD>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int NT = p.HP >> 4, KT = p.KAp >> 4;
  const int tile = idx * 16 + wave;
  if (tile >= KT * NT) return;
  const int mt = tile / NT, nt = tile - mt * NT;
  const int m0 = mt * 16, n0 = nt * 16;
  const int a = p.q & 1, pg = a ^ 1;
  const int slot = (p.q + PIPE_SLOTS - 1) & (PIPE_SLOTS - 1);
  const unsigned slot_bytes = (unsigned)G::ROWS * (unsigned)p.ldxa * 4u;
  const rsrc_t rx = make_rsrc(p.Xa + (size_t)slot * G::ROWS * p.ldxa, slot_bytes);
  const rsrc_t rd = make_rsrc(p.D1 + (size_t)pg * G::MAT, G::MAT * 4u);
  const rsrc_t rw = make_rsrc(p.Wm + (size_t)pg * p.KAp * p.HP, (unsigned)p.KAp * (unsigned)p.HP * 4u);
  const unsigned ldx4 = (unsigned)p.ldxa * 4u;
  const unsigned a_base = (unsigned)(4 * g) * ldx4 + (unsigned)(m0 + fr) * 4u;
  const bool has_d1 = nt < pipe_ntq<TD>(p);           // (the tile of the constant column alone has no gradient: stays zero)
  // up to 8 row blocks: every operand requested before the first MFMA; more: in groups of four (register budget of a
  // 16-wave workgroup)
  constexpr int CH = NRB <= 8 ? NRB : 4;
  float wold[4];
  bld4(rw, pipe_wfrag(mt, nt, NT, lane) * 4u, wold);     // rows m0 + 4g + r, column n0 + fr: one fragment entry
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int r0 = 0; r0 < NRB; r0 += CH) {
    float av[CH][4], bv[CH][4];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int rb = r0 + u;
      const bool ok = has_d1 && rb < NRB;
      bld4(rd, ok ? pfrag<TD>(nt, rb, lane) * 4u : kOOB, bv[u]);
#pragma unroll
      for (int j = 0; j < 4; ++j) av[u][j] = pipe_bld_s(rx, ok ? a_base : kOOB, (unsigned)(16 * rb + j) * ldx4);
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      acc0 = MFMA16(av[u][0], bv[u][0], acc0);
      acc1 = MFMA16(av[u][1], bv[u][1], acc1);
      acc0 = MFMA16(av[u][2], bv[u][2], acc0);
      acc1 = MFMA16(av[u][3], bv[u][3], acc1);
    }
  }
  const int col = n0 + fr;
  PIPE_STAMP(1);
  f32x4 w4;
#pragma unroll
  for (int r = 0; r < 4; ++r) w4[r] = wold[r] - p.lr * (acc0[r] + acc1[r]);
  if (p.flush) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 4 * g + r;
      if (col < p.Hd) {
        if (row < p.In) p.W1[(size_t)row * p.Hd + col] = w4[r];
        else if (row == p.In) p.b1[col] = w4[r];
      }
    }
  } else {
    *reinterpret_cast<f32x4*>(p.Wm + (size_t)a * p.KAp * p.HP + pipe_wfrag(mt, nt, NT, lane)) = w4;
  }
}

// ---------------------------------------------------------------------------------------------------------
// role T: counters, log header, gather of the batch of step s+3, row indices of step s+4
// ---------------------------------------------------------------------------------------------------------
template <int TD>
__device__ __forceinline__ void pipe_role_t(const DtfPipeArgs& p, int tb) {
  // 8 rows per workgroup, 128 lanes per row: a workgroup moves 25 KB in and 25 KB out.  (With 32 rows per workgroup
  // the four gather workgroups were the last of the launch to finish: 100 KB in and out through one CU's 64 B/clk.)
  using G = PipeGeo<TD>;
  const DtfTailArgs& t = p.tail;
  const int tid = threadIdx.x;
  const int a = p.q & 1;
  const int B = p.B, C = p.C;
  const int sub = tid >> 7, l = tid & 127;
  const int i = 8 * tb + sub;
  const int q3 = (p.q + 3) & (PIPE_SLOTS - 1), q4 = (p.q + 4) & (PIPE_SLOTS - 1);
  PIPE_STAMP_CYCLES(2);
  // both chains start here: (row index of step s+3 -> its data) and (counters -> permutation entry of step s+4)
  const int row = i < B ? p.ridx[(size_t)q3 * p.ridx_len + i] : 0;
  const int gs = p.ctr[2 * a], ls = p.ctr[2 * a + 1];
  if (tb == 0 && tid == 0) {
    int* rec = p.wlog + (size_t)DTF_WLOG_INTS * (ls % p.log_cap);
    rec[0] = gs;       // value of global_step before this step's increment
    rec[1] = ls;
    p.ctr[2 * (a ^ 1)] = gs + t.gs_inc;
    p.ctr[2 * (a ^ 1) + 1] = ls + 1;
    t.state[0] = gs + t.gs_inc;
    t.state[1] = ls + 1;
  }
  if (i >= B) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(p.X + (size_t)row * p.ldx);
  float* drow = p.Xa + ((size_t)q3 * G::ROWS + i) * p.ldxa;
  f32x4* dst = reinterpret_cast<f32x4*>(drow);
  const int n4 = p.In >> 2;
  for (int base = 0; base < n4; base += 128 * 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = base + 128 * u + l;
      v[u] = src[k < n4 ? k : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = base + 128 * u + l;
      if (k < n4) dst[k] = v[u];
    }
  }
  if (l == 17) drow[p.In] = 1.f;                       // the column that carries the bias row of W1'
  if (l < 16) {
    const float v = l < C ? p.labels[(size_t)row * C + l] : -__builtin_huge_valf();
    p.labg[q3 * G::LABG + l * G::ROWS + i] = l < C ? v : 0.f;
    // class of the row for the accuracy count: argmax with lowest-index tie break (tf.argmax, distributed.py:83)
    const float vmax = row16_max(v);
    const int am = row16_min(v == vmax ? l : 16);
    if (l == 0) p.labi[q3 * G::ROWS + i] = am;
  }
  if (l == 16) {
    const long long pos64 = (long long)(ls + 4) * B + i;
    const long long epoch = pos64 / t.n_train;
    const int pos = (int)(pos64 - epoch * t.n_train);
    int* rn = p.ridx + (size_t)q4 * p.ridx_len;
    rn[i] = t.perm ? t.perm[(size_t)(epoch & 1) * t.n_train + pos] : pos;
    rn[B + i] = row;                                   // the batch before it: its accuracy-only rows
  }
  PIPE_STAMP(1);
  PIPE_STAMP_CYCLES(3);
}

// ---------------------------------------------------------------------------------------------------------
// output layer brought up to date: element e of W2'(s) = W2m[p][e] - lr * sum over row blocks of P2[p][rb][e]
// ---------------------------------------------------------------------------------------------------------
template <int NRB>
struct PipeW2Loads {
  float w, part[NRB];
};
template <int NRB>
__device__ __forceinline__ void pipe_w2_request(PipeW2Loads<NRB>& o, rsrc_t rw2, rsrc_t rp2, unsigned e4, unsigned blk4) {
  o.w = bld(rw2, e4);
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) o.part[rb] = pipe_bld_s(rp2, e4, (unsigned)rb * blk4);
}
template <int NRB>
__device__ __forceinline__ float pipe_w2_value(const PipeW2Loads<NRB>& o, float lr) {
  float s = 0.f;
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) s += o.part[rb];
  return o.w - lr * s;
}

template <int NRB, int TD>
__device__ __forceinline__ void pipe_role_f(const DtfPipeArgs& p) {
  const int pg = (p.q & 1) ^ 1;
  const unsigned blk4 = (unsigned)p.HP * 64u;
  const rsrc_t rw2 = make_rsrc(p.W2m + (size_t)pg * p.HP * 16, blk4);
  const rsrc_t rp2 = make_rsrc(p.P2 + (size_t)pg * TD * p.HP * 16, (unsigned)TD * blk4);
  for (int e = threadIdx.x; e < p.HP * 16; e += 1024) {
    PipeW2Loads<NRB> ld;
    pipe_w2_request<NRB>(ld, rw2, rp2, (unsigned)e * 4u, blk4);
    const float w = pipe_w2_value<NRB>(ld, p.lr);
    const int ln = (e >> 2) & 63;
    const int k = (e >> 8) * 16 + (ln >> 4) * 4 + (e & 3), c = ln & 15;      // inverse of pipe_w2_off
    if (c < p.C) {
      if (k < p.Hd) p.W2[(size_t)k * p.C + c] = w;
      else if (k == p.Hd) p.b2[c] = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// role L: the dependent path of 16 batch rows
// ---------------------------------------------------------------------------------------------------------
template <int NRB>
__device__ __forceinline__ void pipe_role_l(const DtfPipeArgs& p, int rbl, float* smem) {
  // One wave per 16-column tile of the hidden layer (NT <= 8 waves; the others leave at once).  A wave owns its tile
  // through every stage - the stage-1 product over ALL previous batch rows, its 16 rows of W2', its partial logits,
  // its dz1 tile - so the only exchanges between waves are the logits (sum over the tiles) and dz2 (back to the tiles):
  // two workgroup barriers.  (Splitting K of stage 1 over two waves per tile gains nothing: 7 tiles on 4 SIMDs leave
  // the busiest matrix pipe with 56 MFMAs either way, and the split costs an LDS exchange and a third barrier.)
  // LDS carve-up (floats)
  float (*w2t)[16][17] = reinterpret_cast<float (*)[16][17]>(smem);                    // [8][16][17]  W2'(s) rows of each tile
  float (*hs)[16][17] = reinterpret_cast<float (*)[16][17]>(smem + 2176);             // [8][16][17]  h' tiles
  float (*zp)[16][16] = reinterpret_cast<float (*)[16][16]>(smem + 2176 + 2176);      // [8][16][16]  partial logits
  float (*dzs)[17] = reinterpret_cast<float (*)[17]>(smem + 2176 + 2176 + 2048);      // [16][17]     dz2 of the 16 rows
  float* row_loss = smem + 2176 + 2176 + 2048 + 272;                                  // [16] loss of each row
  int* row_hit = reinterpret_cast<int*>(row_loss + 16);                               // [16] prediction of each row correct

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = p.HP >> 4;
  const int bk_wave = NT > 4 ? NT : 4;                // first wave without a tile or softmax rows: the bookkeeper
  if (wave > bk_wave) return;                         // (waves 0..3 also carry the softmax rows)
  const int fr = lane & 15, g = lane >> 4;
  const bool post = rbl >= NRB;                       // workgroup-uniform
  const int rb = post ? rbl - NRB : rbl;
  const int m0 = rb * 16;
  const int a = p.q & 1, pg = a ^ 1;
  const int tile = wave;
  const bool t_wave = wave < NT;                      // wave-uniform
  const int n0 = tile * 16;
  const int col = n0 + fr;

  // ---- every load of the wave is requested here, before anything waits ----
  // (1) stage 1: z1 tile = Q - lr G D1 over the batch rows of the previous step
  const rsrc_t rg = make_rsrc(p.GG + ((size_t)pg * 3 + (post ? 0 : 1)) * PIPE_MAT, PIPE_MAT * 4u);
  const rsrc_t rd = make_rsrc(p.D1 + (size_t)pg * PIPE_MAT, PIPE_MAT * 4u);
  // training rows start from Q of this step, accuracy-only rows from z1 of the previous step
  const rsrc_t rq = make_rsrc(post ? p.Z1 + (size_t)pg * PIPE_MAT : p.Q + (size_t)a * PIPE_MAT, PIPE_MAT * 4u);
  float ga[NRB][4], db[NRB][4], qv[4];
#pragma unroll
  for (int jb = 0; jb < NRB; ++jb) {
    bld4(rg, t_wave ? pipe_frag(rb, jb, lane) * 4u : kOOB, ga[jb]);
    bld4(rd, t_wave ? pipe_frag(tile, jb, lane) * 4u : kOOB, db[jb]);
  }
  bld4(rq, t_wave ? pipe_frag(rb, tile, lane) * 4u : kOOB, qv);
  // (2) label of (row, class) and class of the row for the softmax threads, the step number for the log record
  //     (buffer loads, no branch around them: a load inside divergent control flow makes the compiler wait for
  //     every outstanding load in front of the first MFMA)
  const int srow = tid >> 4;                          // threads [0,256): row srow, class fr
  const int lslot = post ? ((p.q + PIPE_SLOTS - 1) & (PIPE_SLOTS - 1)) : p.q;
  const rsrc_t rlab = make_rsrc(p.labg + lslot * PIPE_LABG_SLOT, PIPE_LABG_SLOT * 4u);
  const rsrc_t rli = make_rsrc(p.labi + lslot * PIPE_ROWS, PIPE_ROWS * 4u);
  const rsrc_t rctr = make_rsrc(p.ctr, 16u);
  const float lab = bld(rlab, tid < 256 ? (unsigned)(fr * PIPE_ROWS + m0 + srow) * 4u : kOOB);
  const int lab_class = bldi(rli, tid < 256 ? (unsigned)(m0 + srow) * 4u : kOOB);
  const int ls = bldi(rctr, (unsigned)(2 * a + 1) * 4u);
  // (3) the wave's 16 rows of the output layer of this step: lane (class fr, g) gets rows n0 + 4g + j, j = 0..3, of
  //     W2'(s-1) and of every row block's h'^T dz2 in one 16-byte load each
  const unsigned blk4 = (unsigned)p.HP * 64u;
  const rsrc_t rw2 = make_rsrc(p.W2m + (size_t)pg * p.HP * 16, blk4);
  const rsrc_t rp2 = make_rsrc(p.P2 + (size_t)pg * 8 * p.HP * 16, 8u * blk4);
  const unsigned w2off = t_wave ? (unsigned)(tile * 64 + lane) * 16u : kOOB;
  float w2o[4], w2p[NRB][4];
  bld4(rw2, w2off, w2o);
#pragma unroll
  for (int k = 0; k < NRB; ++k) pipe_bld4_s(rp2, w2off, (unsigned)k * blk4, w2p[k]);
  // nothing moves across this line: the scheduler otherwise sinks the operand loads next to their MFMAs and the
  // wave pays one cold round trip per 16-row block instead of one in all
  __builtin_amdgcn_sched_barrier(0);
  PIPE_STAMP(1);

  // ---- stage 1 ----
  float hreg[4] = {0.f, 0.f, 0.f, 0.f}, w2v[4] = {0.f, 0.f, 0.f, 0.f};
  if (t_wave) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb < NRB; ++jb) {
      acc0 = MFMA16(ga[jb][0], db[jb][0], acc0);
      acc1 = MFMA16(ga[jb][1], db[jb][1], acc1);
      acc0 = MFMA16(ga[jb][2], db[jb][2], acc0);
      acc1 = MFMA16(ga[jb][3], db[jb][3], acc1);
    }
    // the output layer brought up to date while the matrix pipe works
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NRB; ++k) sum += w2p[k][j];
      w2v[j] = w2o[j] - p.lr * sum;                    // row n0 + 4g + j, class fr
      w2t[tile][4 * g + j][fr] = w2v[j];
    }
    if (rbl == 0) {                                    // one workgroup keeps W2'(s) for the next launch
      const f32x4 w4 = {w2v[0], w2v[1], w2v[2], w2v[3]};
      *reinterpret_cast<f32x4*>(p.W2m + (size_t)a * p.HP * 16 + (size_t)(tile * 64 + lane) * 4) = w4;
    }
    PIPE_STAMP(2);
    float zv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zv[r] = qv[r] - p.lr * (acc0[r] + acc1[r]);
      // column Hd of h' is the constant 1 that carries b2; columns beyond it do not exist
      hreg[r] = col < p.Hd ? fmaxf(zv[r], 0.f) : (col == p.Hd ? 1.f : 0.f);
      hs[tile][4 * g + r][fr] = hreg[r];
    }
    // ---- stage 2: partial logits of the tile's 16 hidden columns (the wave reads back only its own h tile: LDS
    //      operations of one wave execute in order, no workgroup barrier needed) ----
    __builtin_amdgcn_wave_barrier();
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
    // (MFMA step j takes hidden column n0 + 4g + j: the order in which the output-layer rows sit in the lane)
    z0 = MFMA16(hs[tile][fr][4 * g], w2v[0], z0);
    z1 = MFMA16(hs[tile][fr][4 * g + 1], w2v[1], z1);
    z0 = MFMA16(hs[tile][fr][4 * g + 2], w2v[2], z0);
    z1 = MFMA16(hs[tile][fr][4 * g + 3], w2v[3], z1);
#pragma unroll
    for (int r = 0; r < 4; ++r) zp[tile][4 * g + r][fr] = z0[r] + z1[r];
    if (!post) {                                       // z1 of this batch: the next launch's accuracy-only rows start from it
      const f32x4 z4 = {zv[0], zv[1], zv[2], zv[3]};
      *reinterpret_cast<f32x4*>(p.Z1 + (size_t)a * PIPE_MAT + pipe_frag(rb, tile, lane)) = z4;
    }
  }
  // operands of stages 4 and 5 that do not depend on the softmax are fetched from LDS while the wave would wait
  float w2b[4] = {0.f, 0.f, 0.f, 0.f}, ha[4] = {0.f, 0.f, 0.f, 0.f};
  if (t_wave && !post) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w2b[j] = w2t[tile][fr][4 * j + g];
      ha[j] = hs[tile][4 * j + g][fr];
    }
  }
  __syncthreads();
  PIPE_STAMP(5);
  // ---- stage 3: loss, accuracy, dz2 of the 16 rows (16 lanes per row) ----
  if (tid < 256) {
    float zt[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) zt[t] = zp[t < NT ? t : 0][srow][fr];
    float z = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) z += t < NT ? zt[t] : 0.f;
    const int row = m0 + srow;
    const bool row_ok = row < p.B;
    float loss = 0.f, dz = 0.f;
    int hit = 0;
    pipe_row_head(z, lab, lab_class, fr < p.C, row_ok && !post, fr, p.loss_mode, p.inv_b, loss, hit, dz);
    dzs[srow][fr] = dz;
    if (fr == 0) {                                     // per-row results; added up after the barrier, off the dependent path
      row_loss[srow] = row_ok ? loss : 0.f;
      row_hit[srow] = row_ok ? hit : 0;
    }
  }
  __syncthreads();
  PIPE_STAMP(6);
  if (wave == bk_wave) {
    // the row block's log entries, added up in row order by a wave that has nothing else to do
    if (lane == 0) {
      float l = 0.f;
      int c = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        l += row_loss[r];
        c += row_hit[r];
      }
      int* rec = p.wlog + (size_t)DTF_WLOG_INTS * (ls % p.log_cap);
      if (post) {
        rec[36 + rb] = c;
      } else {
        rec[4 + rb] = __float_as_int(l);
        rec[20 + rb] = c;
      }
    }
    return;
  }
  if (post || !t_wave) return;
  // ---- stage 4: dz1 tile = (dz2 W2^T) masked by the ReLU, kept for the next launch ----
  {
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    d0 = MFMA16(dzs[fr][g], w2b[0], d0);
    d1 = MFMA16(dzs[fr][4 + g], w2b[1], d1);
    d0 = MFMA16(dzs[fr][8 + g], w2b[2], d0);
    d1 = MFMA16(dzs[fr][12 + g], w2b[3], d1);
    f32x4 d4;
#pragma unroll
    for (int r = 0; r < 4; ++r) d4[r] = (col < p.Hd && hreg[r] > 0.f) ? d0[r] + d1[r] : 0.f;
    // B-operand layout: k = batch row m0 + 4g + r, n = col: the four values of a lane are one fragment entry
    *reinterpret_cast<f32x4*>(p.D1 + (size_t)a * PIPE_MAT + pipe_frag(tile, rb, lane)) = d4;
  }
  // ---- stage 5: this row block's share of the output layer's gradient, h'^T dz2 ----
  {
    f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
    u0 = MFMA16(ha[0], dzs[g][fr], u0);
    u1 = MFMA16(ha[1], dzs[4 + g][fr], u1);
    u0 = MFMA16(ha[2], dzs[8 + g][fr], u0);
    u1 = MFMA16(ha[3], dzs[12 + g][fr], u1);
    // result (hidden n0 + 4g + r, class fr): the lane's four values are one entry of the output-layer layout
    const f32x4 u4 = {u0[0] + u1[0], u0[1] + u1[1], u0[2] + u1[2], u0[3] + u1[3]};
    *reinterpret_cast<f32x4*>(p.P2 + ((size_t)a * 8 + rb) * p.HP * 16 + (size_t)(tile * 64 + lane) * 4) = u4;
  }
  PIPE_STAMP(7);
}

// ---------------------------------------------------------------------------------------------------------
// role L for the 16 x 16-tile geometry (batch <= 256, hidden <= 256): the same five stages, written as loops - a
// wave carries tiles wave, wave + 16 of the up to 17 tiles of h', the stage-1 product runs over up to 16 previous
// row blocks, operands of stages 4 and 5 are read back from LDS instead of being kept in registers.
// ---------------------------------------------------------------------------------------------------------
#define PIPE_BIG_SMEM 13904
template <int NRB>
__device__ __forceinline__ void pipe_role_l_big(const DtfPipeArgs& p, int rbl, float* smem) {
  constexpr int TD = 16;
  using G = PipeGeo<TD>;
  float (*w2t)[16][17] = reinterpret_cast<float (*)[16][17]>(smem);                   // [17][16][17]  W2'(s) rows of each tile
  float (*hs)[16][17] = reinterpret_cast<float (*)[16][17]>(smem + 4624);            // [17][16][17]  h' tiles
  float (*zp)[16][16] = reinterpret_cast<float (*)[16][16]>(smem + 9248);            // [17][16][16]  partial logits
  float (*dzs)[17] = reinterpret_cast<float (*)[17]>(smem + 13600);                  // [16][17]      dz2 of the 16 rows
  float* row_loss = smem + 13872;                                                     // [16]
  int* row_hit = reinterpret_cast<int*>(row_loss + 16);                               // [16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = p.HP >> 4;
  const int NTQ = pipe_ntq<TD>(p);
  const int fr = lane & 15, g = lane >> 4;
  const bool post = rbl >= NRB;                       // workgroup-uniform
  const int rb = post ? rbl - NRB : rbl;
  const int m0 = rb * 16;
  const int a = p.q & 1, pg = a ^ 1;

  const rsrc_t rg = make_rsrc(p.GG + ((size_t)pg * 3 + (post ? 0 : 1)) * G::MAT, G::MAT * 4u);
  const rsrc_t rd = make_rsrc(p.D1 + (size_t)pg * G::MAT, G::MAT * 4u);
  const rsrc_t rq = make_rsrc(post ? p.Z1 + (size_t)pg * G::MAT : p.Q + (size_t)a * G::MAT, G::MAT * 4u);
  const int srow = tid >> 4;                          // threads [0,256): row srow, class fr
  const int lslot = post ? ((p.q + PIPE_SLOTS - 1) & (PIPE_SLOTS - 1)) : p.q;
  const rsrc_t rlab = make_rsrc(p.labg + lslot * G::LABG, G::LABG * 4u);
  const rsrc_t rli = make_rsrc(p.labi + lslot * G::ROWS, G::ROWS * 4u);
  const rsrc_t rctr = make_rsrc(p.ctr, 16u);
  const float lab = bld(rlab, tid < 256 ? (unsigned)(fr * G::ROWS + m0 + srow) * 4u : kOOB);
  const int lab_class = bldi(rli, tid < 256 ? (unsigned)(m0 + srow) * 4u : kOOB);
  const int ls = bldi(rctr, (unsigned)(2 * a + 1) * 4u);
  const unsigned blk4 = (unsigned)p.HP * 64u;
  const rsrc_t rw2 = make_rsrc(p.W2m + (size_t)pg * p.HP * 16, blk4);
  const rsrc_t rp2 = make_rsrc(p.P2 + (size_t)pg * TD * p.HP * 16, (unsigned)TD * blk4);

  for (int tile = wave; tile < NT; tile += 16) {      // wave-uniform
    const int n0 = tile * 16, col = n0 + fr;
    const bool has_q = tile < NTQ;                    // (else: the tile of the constant column alone)
    // ---- stage 1: z1 tile = Q - lr G D1 over the batch rows of the previous step ----
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float qv[4];
    bld4(rq, has_q ? pfrag<TD>(rb, tile, lane) * 4u : kOOB, qv);
#pragma unroll 4
    for (int jb = 0; jb < NRB; ++jb) {
      float ga[4], db[4];
      bld4(rg, has_q ? pfrag<TD>(rb, jb, lane) * 4u : kOOB, ga);
      bld4(rd, has_q ? pfrag<TD>(tile, jb, lane) * 4u : kOOB, db);
      acc0 = MFMA16(ga[0], db[0], acc0);
      acc1 = MFMA16(ga[1], db[1], acc1);
      acc0 = MFMA16(ga[2], db[2], acc0);
      acc1 = MFMA16(ga[3], db[3], acc1);
    }
    // the tile's 16 rows of the output layer brought up to date
    const unsigned w2off = (unsigned)(tile * 64 + lane) * 16u;
    float w2o[4], w2s[4] = {0.f, 0.f, 0.f, 0.f}, w2v[4];
    bld4(rw2, w2off, w2o);
#pragma unroll 4
    for (int k = 0; k < NRB; ++k) {
      float part[4];
      pipe_bld4_s(rp2, w2off, (unsigned)k * blk4, part);
#pragma unroll
      for (int j = 0; j < 4; ++j) w2s[j] += part[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w2v[j] = w2o[j] - p.lr * w2s[j];                 // row n0 + 4g + j, class fr
      w2t[tile][4 * g + j][fr] = w2v[j];
    }
    if (rbl == 0) {                                    // one workgroup keeps W2'(s) for the next launch
      const f32x4 w4 = {w2v[0], w2v[1], w2v[2], w2v[3]};
      *reinterpret_cast<f32x4*>(p.W2m + (size_t)a * p.HP * 16 + (size_t)(tile * 64 + lane) * 4) = w4;
    }
    float zv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zv[r] = qv[r] - p.lr * (acc0[r] + acc1[r]);
      hs[tile][4 * g + r][fr] = col < p.Hd ? fmaxf(zv[r], 0.f) : (col == p.Hd ? 1.f : 0.f);
    }
    // ---- stage 2: partial logits of the tile's 16 hidden columns ----
    __builtin_amdgcn_wave_barrier();
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
    z0 = MFMA16(hs[tile][fr][4 * g], w2v[0], z0);
    z1 = MFMA16(hs[tile][fr][4 * g + 1], w2v[1], z1);
    z0 = MFMA16(hs[tile][fr][4 * g + 2], w2v[2], z0);
    z1 = MFMA16(hs[tile][fr][4 * g + 3], w2v[3], z1);
#pragma unroll
    for (int r = 0; r < 4; ++r) zp[tile][4 * g + r][fr] = z0[r] + z1[r];
    if (!post && has_q) {
      const f32x4 z4 = {zv[0], zv[1], zv[2], zv[3]};
      *reinterpret_cast<f32x4*>(p.Z1 + (size_t)a * G::MAT + pfrag<TD>(rb, tile, lane)) = z4;
    }
  }
  __syncthreads();
  // ---- stage 3: loss, accuracy, dz2 of the 16 rows (16 lanes per row) ----
  if (tid < 256) {
    float z = 0.f;
    for (int t = 0; t < NT; ++t) z += zp[t][srow][fr];
    const int row = m0 + srow;
    const bool row_ok = row < p.B;
    float loss = 0.f, dz = 0.f;
    int hit = 0;
    pipe_row_head(z, lab, lab_class, fr < p.C, row_ok && !post, fr, p.loss_mode, p.inv_b, loss, hit, dz);
    dzs[srow][fr] = dz;
    if (fr == 0) {
      row_loss[srow] = row_ok ? loss : 0.f;
      row_hit[srow] = row_ok ? hit : 0;
    }
  }
  __syncthreads();
  if (wave == 15 && lane == 0) {                       // the row block's log entries, added up in row order
    float l = 0.f;
    int c = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      l += row_loss[r];
      c += row_hit[r];
    }
    int* rec = p.wlog + (size_t)DTF_WLOG_INTS * (ls % p.log_cap);
    if (post) {
      rec[36 + rb] = c;
    } else {
      rec[4 + rb] = __float_as_int(l);
      rec[20 + rb] = c;
    }
  }
  if (post) return;
  for (int tile = wave; tile < NT; tile += 16) {
    const int n0 = tile * 16, col = n0 + fr;
    float w2b[4], ha[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w2b[j] = w2t[tile][fr][4 * j + g];
      ha[j] = hs[tile][4 * j + g][fr];
    }
    // ---- stage 4: dz1 tile = (dz2 W2^T) masked by the ReLU, kept for the next launch ----
    if (tile < NTQ) {
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = MFMA16(dzs[fr][g], w2b[0], d0);
      d1 = MFMA16(dzs[fr][4 + g], w2b[1], d1);
      d0 = MFMA16(dzs[fr][8 + g], w2b[2], d0);
      d1 = MFMA16(dzs[fr][12 + g], w2b[3], d1);
      f32x4 d4;
#pragma unroll
      for (int r = 0; r < 4; ++r) d4[r] = (col < p.Hd && hs[tile][4 * g + r][fr] > 0.f) ? d0[r] + d1[r] : 0.f;
      *reinterpret_cast<f32x4*>(p.D1 + (size_t)a * G::MAT + pfrag<TD>(tile, rb, lane)) = d4;
    }
    // ---- stage 5: this row block's share of the output layer's gradient, h'^T dz2 ----
    {
      f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
      u0 = MFMA16(ha[0], dzs[g][fr], u0);
      u1 = MFMA16(ha[1], dzs[4 + g][fr], u1);
      u0 = MFMA16(ha[2], dzs[8 + g][fr], u0);
      u1 = MFMA16(ha[3], dzs[12 + g][fr], u1);
      const f32x4 u4 = {u0[0] + u1[0], u0[1] + u1[1], u0[2] + u1[2], u0[3] + u1[3]};
      *reinterpret_cast<f32x4*>(p.P2 + ((size_t)a * TD + rb) * p.HP * 16 + (size_t)(tile * 64 + lane) * 4) = u4;
    }
  }
}

// Role C (prime launch): both generations of the working copies from the flat parameter buffer, no pending gradient
// (D1, P2 zero), counters of the first step.  Workgroup `cb` handles 4096 elements.
template <int TD>
__device__ __forceinline__ void pipe_role_c(const DtfPipeArgs& p, int cb) {
  const int nWm = p.KAp * p.HP, nW2 = p.HP * 16, nD1 = PipeGeo<TD>::MAT, nP2 = TD * p.HP * 16;
  const int total = 2 * (nWm + nW2 + nD1 + nP2);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int e = cb * 4096 + u * 1024 + (int)threadIdx.x;
    if (e >= total) break;
    if (e < 2 * nWm) {
      const int w = e >= nWm ? e - nWm : e;
      // inverse of pipe_wfrag: w = ((kt * NT + nt) * 64 + lane) * 4 + (k % 4)
      const int NTc = p.HP >> 4;
      const int tl = w >> 8, ln = (w >> 2) & 63;
      const int k = (tl / NTc) * 16 + (ln >> 4) * 4 + (w & 3), c = (tl % NTc) * 16 + (ln & 15);
      float v = 0.f;
      if (c < p.Hd) {
        if (k < p.In) v = p.W1[(size_t)k * p.Hd + c];
        else if (k == p.In) v = p.b1[c];
      }
      p.Wm[e] = v;
      continue;
    }
    e -= 2 * nWm;
    if (e < 2 * nW2) {
      const int w = e >= nW2 ? e - nW2 : e;
      const int ln = (w >> 2) & 63;
      const int k = (w >> 8) * 16 + (ln >> 4) * 4 + (w & 3), c = ln & 15;    // inverse of pipe_w2_off
      float v = 0.f;
      if (c < p.C) {
        if (k < p.Hd) v = p.W2[(size_t)k * p.C + c];
        else if (k == p.Hd) v = p.b2[c];
      }
      p.W2m[e] = v;
      continue;
    }
    e -= 2 * nW2;
    if (e < 2 * nD1) { p.D1[e] = 0.f; continue; }
    e -= 2 * nD1;
    p.P2[e] = 0.f;
  }
  if (cb == 0 && threadIdx.x < 2) {
    // p.q is the slot of the launch BEFORE the first step: that step reads generation (q + 1) & 1 of the counters
    p.ctr[2 * ((p.q & 1) ^ 1) + threadIdx.x] = p.tail.state[threadIdx.x];
  }
}

template <int NRB, int TD>
__global__ __launch_bounds__(1024) void pipe_step_kernel(DtfPipeArgs p) {
  __shared__ float smem[TD == 16 ? PIPE_BIG_SMEM : 9216];
  const PipeDims d = pipe_dims<TD>(p, NRB);
  // Which XCD a workgroup runs on decides what its loads find in that XCD's L2: workgroup i of a launch goes to XCD
  // i mod 8, and a kernel boundary leaves an L2 with what ITS workgroups read and wrote.  While the whole grid is
  // resident at once (<= 512 workgroups) the role index is therefore taken XCD-major (blocks 0, 8, 16, ... get indices
  // 0, 1, 2, ...): the row workgroups - producers and consumers of dz1, the output-layer shares, z1 and W2' from launch
  // to launch - all sit on XCD 0, the Gram tiles that share batch rows sit together on the next XCDs, and so on.
  // Measured on the reference shape (2000 steps, one box, us/step): plain block order 6.42; XCD-major with the roles in
  // the order L G Q W T 5.69, L Q G W T 5.69, L G T W Q 5.70, L T W Q G 6.19, L T G Q W 6.22, L G W Q T 6.25 (the
  // gather or the weight tiles next to the row workgroups, or all weight tiles on one XCD, cost what the grouping
  // wins); row workgroups on XCD 0 and everything else dealt round-robin over the other seven 6.10; row workgroups
  // left spread and only the rest grouped 6.38.  Larger grids run in waves of workgroups; there block order is
  // dispatch order and the roles with the longest dependent chains go first (256 / 256: 21.1 against 24.5 us).
  const bool xcd_major = gridDim.x <= 512;
  int b = xcd_major ? dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  PIPE_STAMP(0);
  if (b < d.nL) {
    if constexpr (TD == 16) pipe_role_l_big<NRB>(p, b, smem);
    else pipe_role_l<NRB>(p, b, smem);
    return;
  }
  b -= d.nL;
  if (xcd_major) {
    if (b < d.nG) { pipe_role_g<NRB, TD>(p, b, smem); return; }
    b -= d.nG;
    if (b < d.nQ) { pipe_role_q<NRB, TD>(p, b, smem); return; }
    b -= d.nQ;
    if (b < d.nW) { pipe_role_w<NRB, TD>(p, b); return; }
    b -= d.nW;
    if (b < d.nT) { pipe_role_t<TD>(p, b); return; }
    b -= d.nT;
  } else {
    if (b < d.nT) { pipe_role_t<TD>(p, b); return; }
    b -= d.nT;
    if (b < d.nW) { pipe_role_w<NRB, TD>(p, b); return; }
    b -= d.nW;
    if (b < d.nQ) { pipe_role_q<NRB, TD>(p, b, smem); return; }
    b -= d.nQ;
    if (b < d.nG) { pipe_role_g<NRB, TD>(p, b, smem); return; }
    b -= d.nG;
  }
  if (b < d.nF) { pipe_role_f<NRB, TD>(p); return; }
  b -= d.nF;
  if (b < d.nC) pipe_role_c<TD>(p, b);
}

// geometry of a shape: 8 x 8 tiles while batch and hidden + 1 fit 128, else 16 x 16
static int pipe_td_of(int hd, int b) { return (hd <= 127 && b <= 128) ? 8 : 16; }

static int pipe_validate(const DtfPipeArgs* a) {
  if (!a || !a->X || !a->labels || !a->Xa || !a->labg || !a->ridx || !a->Wm || !a->W2m || !a->Q || !a->GG || !a->D1 || !a->P2 ||
      !a->ctr || !a->Z1 || !a->labi || !a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->wlog || !a->tail.state)
    return DTF_ERR_ARG;
  if (a->In <= 0 || (a->In & 3) || a->Hd <= 0 || a->Hd > 256 || a->C <= 0 || a->C > 16 || a->B <= 0 || a->B > 256)
    return DTF_ERR_SHAPE;
  if (a->td != pipe_td_of(a->Hd, a->B)) return DTF_ERR_SHAPE;                  // the buffers were laid out for this geometry
  const int rows = 16 * a->td, nch = a->td == 16 ? 5 : 4;
  const long long mat = 256ll * a->td * a->td;
  if (a->R != a->B && a->R != 2 * a->B) return DTF_ERR_SHAPE;
  if (a->HP != ((a->Hd + 1 + 15) & ~15) || a->KAp != ((a->In + 1 + 15) & ~15) || a->ldxa != a->KAp) return DTF_ERR_SHAPE;
  if ((a->KAp >> 4) + dtf_cdiv(a->B, 16) > 16 * nch) return DTF_ERR_SHAPE;     // split-K of role Q: 16 waves x nch chunks
  if ((a->KAp >> 4) > 64) return DTF_ERR_SHAPE;                                // split-K of role G: 16 waves x 4 chunks
  if (a->q < 0 || a->q >= PIPE_SLOTS || a->ldx < a->In || (a->ldx & 3) || a->log_cap <= 0 || a->ridx_len < 2 * a->B)
    return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->X) || !dtf_aligned16(a->Xa) || !dtf_aligned16(a->GG) || !dtf_aligned16(a->D1) || !dtf_aligned16(a->Q) || !dtf_aligned16(a->Z1)) return DTF_ERR_ALIGN;
  if ((uint64_t)rows * a->ldxa * 4u >= 0xFFFFFFF0u || (uint64_t)a->KAp * a->HP * 4u >= 0xFFFFFFF0u) return DTF_ERR_SHAPE;
  if (a->tail.n_train <= 0 || !a->tail.enable) return DTF_ERR_SHAPE;
  if (a->roles <= 0 || a->roles > 127) return DTF_ERR_ARG;
  if (a->flush && (a->roles & ~(PIPE_ROLE_W | PIPE_ROLE_F))) return DTF_ERR_ARG;
  if (a->prime && (a->flush || a->roles != (PIPE_ROLE_Q | PIPE_ROLE_C))) return DTF_ERR_ARG;
  if ((a->roles & PIPE_ROLE_C) && !a->prime) return DTF_ERR_ARG;
  if (2ll * ((long long)a->KAp * a->HP + mat + (1ll + a->td) * a->HP * 16) >= 0x7fffffffll) return DTF_ERR_SHAPE;
  return DTF_OK;
}

template <int NRB, int TD>
static void pipe_launch(const DtfPipeArgs& a, hipStream_t s) {
  const PipeDims d = pipe_dims<TD>(a, NRB);
  const int nblk = d.nL + d.nQ + d.nG + d.nW + d.nT + d.nF + d.nC;
  hipLaunchKernelGGL((pipe_step_kernel<NRB, TD>), dim3(nblk), dim3(1024), 0, s, a);
}

static int pipe_launch_any(const DtfPipeArgs& a, hipStream_t s) {
  const int nrb = dtf_cdiv(a.B, 16);
  if (a.td == 8) {
    if (nrb <= 2) pipe_launch<2, 8>(a, s);
    else if (nrb <= 4) pipe_launch<4, 8>(a, s);
    else if (nrb <= 7) pipe_launch<7, 8>(a, s);
    else pipe_launch<8, 8>(a, s);
  } else {
    if (nrb <= 4) pipe_launch<4, 16>(a, s);
    else if (nrb <= 8) pipe_launch<8, 16>(a, s);
    else pipe_launch<16, 16>(a, s);       // (13 row blocks - batch 200 - as its own instance spilled registers)
  }
  return dtf_check_launch();
}

// One step: every role.  `a->q` = slot of the step.
DTF_EXPORT int dtf_pipe_step(const DtfPipeArgs* a, hipStream_t s) {
  const int rc = pipe_validate(a);
  if (rc != DTF_OK) return rc;
  if (a->flush || a->prime || a->roles != (PIPE_ROLE_L | PIPE_ROLE_Q | PIPE_ROLE_G | PIPE_ROLE_W | PIPE_ROLE_T)) return DTF_ERR_ARG;
  return pipe_launch_any(*a, s);
}

// Start of a run whose first step has slot (a->q + 1) & 7: working copies from the flat buffer and Q of that step
// (products with the flat parameters themselves), in one launch.
DTF_EXPORT int dtf_pipe_prime(const DtfPipeArgs* a, hipStream_t s) {
  const int rc = pipe_validate(a);
  if (rc != DTF_OK) return rc;
  if (a->flush || !a->prime || a->roles != (PIPE_ROLE_Q | PIPE_ROLE_C)) return DTF_ERR_ARG;
  return pipe_launch_any(*a, s);
}

// End of a run whose last step had slot (a->q - 1) & 7: the parameters after that step into the flat buffer.
DTF_EXPORT int dtf_pipe_flush(const DtfPipeArgs* a, hipStream_t s) {
  const int rc = pipe_validate(a);
  if (rc != DTF_OK) return rc;
  if (!a->flush || a->prime || a->roles != (PIPE_ROLE_W | PIPE_ROLE_F)) return DTF_ERR_ARG;
  return pipe_launch_any(*a, s);
}

// Measurement only (tools/pipe_probe.py): any subset of the step roles in one launch, results meaningless.
DTF_EXPORT int dtf_pipe_launch_roles(const DtfPipeArgs* a, hipStream_t s) {
  const int rc = pipe_validate(a);
  if (rc != DTF_OK) return rc;
  return pipe_launch_any(*a, s);
}
