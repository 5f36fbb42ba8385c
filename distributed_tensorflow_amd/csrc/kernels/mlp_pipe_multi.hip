// One-launch training step of the reference network for N SYNCHRONOUS workers of one node (one process per GPU),
// SyncReplicasOptimizer with replicas_to_aggregate == num_workers: every worker computes the gradient of its own
// batch on the same parameters, the mean is applied once (/root/reference/distributed.py:91-102).
//
// The one-worker pipelined step (mlp_pipe.hip) never waits for W1 to be materialised: its dependent path works on
// products with weights of two steps ago plus Gram-matrix corrections.  The same algebra over the batches of ALL
// workers (lrN = lr / N, X'_r(s) = batch of worker r at step s with a column of ones, D_r(s) = its dz1):
//
//     W(s+1) = W(s) - lrN sum_r X'_r(s)^T D_r(s)
//     z1_r(s) = X'_r(s) W(s-2)  -  lrN sum_r' G_rr'(s,s-2) D_r'(s-2)  -  lrN sum_r' G_rr'(s,s-1) D_r'(s-1)
//               `-- Q_r(s): role Q, launch s-1 --'  `-- complete when launch s starts --'  `-- ARRIVES during launch s --'
//
// with G_rr'(a,b) = X'_r(a) X'_r'(b)^T.  Every rank holds the whole data set and every rank's shuffle, so it gathers
// every worker's batches itself; what crosses xGMI per step and rank is D_r (B x H, 40 KB) and the rank's share of
// the output layer's gradient (per 16-row block h'^T dz2: 7 x 4 KB) instead of the weight gradient (318 KB), in ONE
// hop: a row workgroup stores its results straight into the boxes of every rank (own rank included).  No flag and no
// wait for an acknowledgement: every 32-bit word of a box carries its own arrival mark - the lowest mantissa bit is
// overwritten with the PHASE of the step that wrote it, (step >> 2) & 1, which flips every time a box generation is
// reused - so a reader requests the boxes of step s-1 together with its other operands, looks at the marks of what
// came back and asks again only if a word still shows the phase of four steps ago.  (One round trip to uncached
// memory on the dependent path of a step instead of three: store acknowledged -> flag -> flag seen -> boxes read.
// Every rank reads the same bits, marks included, so the replicas stay bit-identical; the exchanged values lose their
// last mantissa bit.)
//
// Only the row workgroups (14 per rank) ever wait.  The weight update and the Q products of launch s use D(s-2),
// which every peer delivered before launch s-1 could end; so a rank's working copy of W1' lags two steps (one more
// than with one worker) and the row workgroups apply two Gram corrections, the first of which (D(s-2)) needs nothing
// of this launch.  Boxes rotate over four generations (step & 3): a peer can be one launch ahead (it needs this
// rank's D(s) for more), and the update / Q roles of this rank's launch s still read generation s-2 then.  (The last
// 4 KB of a region, once flags, are unused.)
//
// Launch of step s, slot q = s mod 8, a = s & 1, p = 1 - a, boxes g0 = s & 3 (written), g1 = (s-1) & 3, g2 = (s-2) & 3:
//   role L  row workgroups (training row blocks, then accuracy-only row blocks).  Waves 0..NT-1 own one 16-column
//           tile of the hidden layer each; waves 8..8+NT-1 bring the output layer up to date for the same tiles
//           (U(s) = U(s-1) - lrN sum over ranks and row blocks of the boxes' h'^T dz2) while the tile waves run the
//           corrections.  The Gram strips of the row block (A operands, the same for every tile) are staged once per
//           workgroup in LDS by LDS-DMA.  Reads Q[a] (accuracy-only rows: Z1[p]), GG[p], boxes g2 and g1, W2m[p];
//           writes boxes g0 of EVERY rank, Z1[a], W2m[a], the log record.
//   role Q  Q[p] (for step s+1) = X'_r(s+1) Wm[p] - lrN sum_r' G_rr'(s+1,s-2) D_r'(s-2)
//   role G  GG[a] = { G(s,s), G(s+1,s), G(s+1,s-1), G(s+2,s-1) } against every rank's batches: all read by launch s+1
//   role W  Wm[a] = Wm[p] - lrN sum_r' X'_r'(s-2)^T D_r'(s-2)   (flush: both pending updates, into W1 / b1)
//   role T  counters, log header; gather of every rank's batch of step s+3; row indices of step s+4
//   role F  (flush) W2, b2 of the flat buffer; the flush launch runs behind a one-workgroup launch that waits for the
//           last step's boxes, so that no more than the row workgroups ever spin on a GPU that ranks share
//   role C  (prime) working copies from the flat parameters, this rank's boxes zeroed and marked (the generation the
//           first step reads as "step before it": arrived; the others: not yet); the host lets no rank start a step
//           before every rank has primed
#include "mlp_pipe_common.h"

typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

// Diagnostic build only (-DDTF_STAMPS): timestamps of the 100 MHz constant clock per row workgroup (index rbl).
#ifdef DTF_STAMPS
__device__ unsigned long long* pipem_stamp_buf = nullptr;
#define PM_STAMP(slot)                                                                                                           \
  do {                                                                                                                           \
    if (threadIdx.x == 0 && pipem_stamp_buf) pipem_stamp_buf[(size_t)rbl * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int dtf_set_stamp_buffer_pipem(unsigned long long* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(pipem_stamp_buf), &p, sizeof(p));
}
#else
#define PM_STAMP(slot) do { } while (0)
#endif

#define PM_GENS 4
#define PM_MAXR 8
#define PM_P2_BLK 2048                                   // floats per (generation, source, row block) of the output-layer boxes
#define PM_P2_OFF ((size_t)PM_GENS * PM_MAXR * PIPE_MAT * 4)
#define PM_FLAG_OFF (PM_P2_OFF + (size_t)PM_GENS * PM_MAXR * 8 * PM_P2_BLK * 4)
#define PM_SMEM_BASE 9216                                // floats of LDS in front of the Gram strips
static_assert(PM_FLAG_OFF + 4096 == DTF_PIPEM_REGION_BYTES, "region layout");

__device__ __forceinline__ float* pm_d1(char* region, int gen, int src) {
  return reinterpret_cast<float*>(region) + ((size_t)gen * PM_MAXR + src) * PIPE_MAT;
}
__device__ __forceinline__ float* pm_p2(char* region, int gen, int src, int rb) {
  return reinterpret_cast<float*>(region + PM_P2_OFF) + (((size_t)gen * PM_MAXR + src) * 8 + rb) * PM_P2_BLK;
}
__device__ __forceinline__ const float* pm_xa(const DtfPipeMultiArgs& m, int slot, int src) {
  return m.b.Xa + ((size_t)(slot & (PIPE_SLOTS - 1)) * m.world + src) * PIPE_ROWS * m.b.ldxa;
}
__device__ __forceinline__ float* pm_gg(const DtfPipeMultiArgs& m, int gen, int kind, int src) {
  return m.b.GG + (((size_t)gen * 4 + kind) * m.world + src) * PIPE_MAT;
}
__device__ __forceinline__ int* pm_ridx(const DtfPipeMultiArgs& m, int slot, int src) {
  return m.b.ridx + ((size_t)(slot & (PIPE_SLOTS - 1)) * m.world + src) * m.b.ridx_len;
}

// 16-byte load of box contents: system scope (sc0 sc1), served by memory whatever a cache may still hold of the same
// address from four steps ago.  The boxes live in uncached memory; with every read of them marked like this no cache
// invalidation (an acquire fence costs one per waiting wave) is needed between seeing a flag and reading the boxes.
__device__ __forceinline__ void pm_bld4_sys(rsrc_t r, unsigned voff, unsigned soff, float (&v)[4]) {
  auto t = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 17);
  v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
}

// 16-byte store that goes straight to memory, wherever it lives (own region or a peer's over xGMI)
__device__ __forceinline__ void pm_store4_sys(rsrc_t r, unsigned off, const f32x4& v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)off, 0, 17);      // sc0 sc1: system scope
}

struct PmDims {
  int NT, KT, nL, nQ, nG, nW, nT, nF, nC, ks, tpw;
};
__host__ __device__ __forceinline__ int pm_ksplit(int world) { return world <= 1 ? 1 : world == 2 ? 2 : world <= 4 ? 4 : 8; }
__host__ __device__ __forceinline__ long long pm_copy_elems(const DtfPipeMultiArgs& m) {
  const DtfPipeArgs& p = m.b;
  return 2ll * ((long long)p.KAp * p.HP + (long long)p.HP * 16) + (long long)m.world * (PM_GENS * (PIPE_MAT + 8 * PM_P2_BLK));
}
__host__ __device__ __forceinline__ PmDims pm_dims(const DtfPipeMultiArgs& m, int nrb) {
  const DtfPipeArgs& p = m.b;
  PmDims d;
  const int post = p.R > p.B ? 1 : 0;
  d.NT = p.HP >> 4;
  d.KT = p.KAp >> 4;
  d.ks = pm_ksplit(m.world);
  d.tpw = 16 / d.ks;
  d.nL = (p.roles & PIPE_ROLE_L) ? nrb * (1 + post) : 0;
  d.nQ = (p.roles & PIPE_ROLE_Q) ? nrb * (1 + (p.prime ? post : 0)) * d.NT : 0;
  d.nG = (p.roles & PIPE_ROLE_G) ? 4 * m.world * nrb * ((nrb + 1) / 2) : 0;
  d.nW = (p.roles & PIPE_ROLE_W) ? (d.KT * d.NT + d.tpw - 1) / d.tpw : 0;
  d.nT = (p.roles & PIPE_ROLE_T) ? m.world * ((p.B + 7) >> 3) : 0;
  d.nF = (p.roles & PIPE_ROLE_F) ? 1 : 0;
  d.nC = (p.roles & PIPE_ROLE_C) ? (int)((pm_copy_elems(m) + 4095) / 4096) : 0;
  return d;
}

// ---------------------------------------------------------------------------------------------------------
// arrival marks: bit 0 of every word of a box = phase of the step that wrote it
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pm_phase(int step) { return (step >> 2) & 1; }      // (arithmetic shift: step -1 has phase 1)
__device__ __forceinline__ float pm_mark(float v, int phase) {
  return __uint_as_float((__float_as_uint(v) & ~1u) | (unsigned)phase);
}
// 0 if the four words carry `phase`
__device__ __forceinline__ unsigned pm_stale4(const float (&v)[4], int phase) {
  const unsigned all_or = __float_as_uint(v[0]) | __float_as_uint(v[1]) | __float_as_uint(v[2]) | __float_as_uint(v[3]);
  const unsigned all_and = __float_as_uint(v[0]) & __float_as_uint(v[1]) & __float_as_uint(v[2]) & __float_as_uint(v[3]);
  return phase ? (~all_and & 1u) : (all_or & 1u);
}
// what the prime launch leaves in a box word of generation `gen` when the first step of the run is step `first`
__device__ __forceinline__ float pm_prime_word(int first, int gen) {
  const int before = first - 1;
  if (gen == (before & 3)) return __uint_as_float((unsigned)pm_phase(before));
  const int next_writer = first + ((gen - first) & 3);
  return __uint_as_float((unsigned)(pm_phase(next_writer) ^ 1));
}
// One decision per wave after a look at `stale` (words that do not show the phase yet, per lane): PM_ARRIVED, PM_AGAIN
// (after a short sleep), or PM_GAVE_UP after the time limit (or at once when the transport has failed earlier in the
// run): the step then goes on with what is there and the host sees the error word.
enum { PM_ARRIVED = 0, PM_AGAIN = 1, PM_GAVE_UP = 2 };
struct PmPoll {
  unsigned long long t0;
  bool started, dead;
};
__device__ __forceinline__ int pm_poll(const DtfPipeMultiArgs& m, PmPoll& st, unsigned stale, int lane) {
  if (__builtin_amdgcn_ballot_w64(stale != 0) == 0) return PM_ARRIVED;
  if (!st.started) {
    st.started = true;
    st.dead = __hip_atomic_load(m.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    st.t0 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long now = __builtin_amdgcn_readfirstlane((unsigned)(__builtin_amdgcn_s_memrealtime() - st.t0));
  if (st.dead || now > m.timeout_ticks) {
    if (lane == 0) __hip_atomic_store(m.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return PM_GAVE_UP;
  }
  __builtin_amdgcn_s_sleep(2);
  return PM_AGAIN;
}
// Waiting is done on ONE 1 KB piece (16 bytes per lane at `voff` + `soff` of `r`): a wave that asked again for all of
// its operands in a loop would keep tens of kilobytes of uncached reads in flight per look, times every wave of every
// row workgroup - traffic the peers it is waiting for have to compete with when ranks share a GPU.
__device__ __forceinline__ int pm_wait_piece(const DtfPipeMultiArgs& m, PmPoll& st, rsrc_t r, unsigned voff, unsigned soff, int phase,
                                             int lane) {
  for (;;) {
    float v[4];
    pm_bld4_sys(r, voff, soff, v);
    const int rc = pm_poll(m, st, pm_stale4(v, phase), lane);
    if (rc != PM_AGAIN) return rc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// role L
// ---------------------------------------------------------------------------------------------------------
// B operands of one correction for two source ranks: the D tiles (hidden tile `tile`, every row block of the source)
template <int NRB>
__device__ __forceinline__ void pm_load_d(const DtfPipeMultiArgs& m, int gen, int s0, int tile, bool active, int lane,
                                          float (&db)[2][NRB][4]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int src = s0 + u;                            // wave-uniform
    const bool on = active && src < m.world;
    const rsrc_t rd = make_rsrc(pm_d1(m.region[m.rank], gen, on ? src : 0), PIPE_MAT * 4u);
#pragma unroll
    for (int jb = 0; jb < NRB; ++jb) pm_bld4_sys(rd, on ? pipe_frag(tile, jb, lane) * 4u : kOOB, 0, db[u][jb]);
  }
}
// index (u * NRB + jb, wave-uniform) of the first loaded tile with a word that does not show `phase`, or -1
template <int NRB>
__device__ __forceinline__ int pm_first_stale(const DtfPipeMultiArgs& m, int s0, int phase, const float (&db)[2][NRB][4]) {
  int bad = -1;
#pragma unroll
  for (int u = 1; u >= 0; --u) {
    if (s0 + u >= m.world) continue;
#pragma unroll
    for (int jb = NRB - 1; jb >= 0; --jb)
      if (__builtin_amdgcn_ballot_w64(pm_stale4(db[u][jb], phase) != 0) != 0) bad = u * NRB + jb;
  }
  return bad;
}
// the D tiles of one step that may still be on their way: what the request brought is looked at; while a tile is
// incomplete the wave waits on that tile alone and then asks for everything again
template <int NRB>
__device__ __forceinline__ void pm_await_d(const DtfPipeMultiArgs& m, int gen, int s0, int tile, int lane, int phase, PmPoll& st,
                                           float (&db)[2][NRB][4]) {
  for (;;) {
    const int bad = pm_first_stale<NRB>(m, s0, phase, db);
    if (bad < 0) return;
    const int u = bad / NRB, jb = bad - u * NRB;
    const rsrc_t rd = make_rsrc(pm_d1(m.region[m.rank], gen, s0 + u), PIPE_MAT * 4u);
    if (pm_wait_piece(m, st, rd, pipe_frag(tile, jb, lane) * 4u, 0, phase, lane) == PM_GAVE_UP) return;
    pm_load_d<NRB>(m, gen, s0, tile, true, lane, db);
  }
}
// acc += strip(kind slot kk, source) . D tiles, A operands from the LDS strips
template <int NRB>
__device__ __forceinline__ void pm_correct(const DtfPipeMultiArgs& m, const float* strips, int kk, int s0, int lane,
                                           const float (&db)[2][NRB][4], f32x4& acc0, f32x4& acc1) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int src = s0 + u;
    if (src >= m.world) break;                         // (nothing staged for it: LDS there is not even initialised)
#pragma unroll
    for (int jb = 0; jb < NRB; ++jb) {
      const f32x4 ga = *reinterpret_cast<const f32x4*>(strips + ((size_t)(kk * m.world + src) * NRB + jb) * 256 + lane * 4);
      acc0 = MFMA16(ga[0], db[u][jb][0], acc0);
      acc1 = MFMA16(ga[1], db[u][jb][1], acc1);
      acc0 = MFMA16(ga[2], db[u][jb][2], acc0);
      acc1 = MFMA16(ga[3], db[u][jb][3], acc1);
    }
  }
}

template <int NRB>
__device__ __forceinline__ void pm_role_l(const DtfPipeMultiArgs& m, int rbl, float* smem) {
  const DtfPipeArgs& p = m.b;
  float (*w2t)[16][17] = reinterpret_cast<float (*)[16][17]>(smem);                    // [8][16][17]  W2'(s) rows of each tile
  float (*hs)[16][17] = reinterpret_cast<float (*)[16][17]>(smem + 2176);             // [8][16][17]  h' tiles
  float (*zp)[16][16] = reinterpret_cast<float (*)[16][16]>(smem + 2176 + 2176);      // [8][16][16]  partial logits
  float (*dzs)[17] = reinterpret_cast<float (*)[17]>(smem + 2176 + 2176 + 2048);      // [16][17]     dz2 of the 16 rows
  float* row_loss = smem + 2176 + 2176 + 2048 + 272;                                  // [16]
  int* row_hit = reinterpret_cast<int*>(row_loss + 16);                               // [16]
  float (*acca)[4][64] = reinterpret_cast<float (*)[4][64]>(smem + 6704);             // [8][4][64]  D(s-2) term of each tile
  float* strips = smem + PM_SMEM_BASE;                                                // [kinds][world][NRB][256]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NT = p.HP >> 4;
  const int world = m.world;
  const int fr = lane & 15, g = lane >> 4;
  const bool post = rbl >= NRB;                       // workgroup-uniform
  const int rb = post ? rbl - NRB : rbl;
  const int m0 = rb * 16;
  const int a = p.q & 1, pg = a ^ 1;
  const int g0 = p.q & 3, g1 = (p.q + 3) & 3, g2 = (p.q + 2) & 3;
  char* const mine = m.region[m.rank];
  const bool t_wave = wave < NT;                      // owns hidden tile `wave`
  const bool o_wave = wave >= 8 && wave < 8 + NT;     // brings the output-layer rows of tile `wave - 8` up to date
  const int tile = t_wave ? wave : (o_wave ? wave - 8 : 0);
  const int n0 = tile * 16;
  const int col = n0 + fr;
  PM_STAMP(0);

  // ---- stage 0: the Gram strips of this row block into LDS, all 16 waves (training rows: against D(s-2) then against
  //      D(s-1); accuracy-only rows: G(s-1,s-1) against D(s-1)) ----
  const int nk = post ? 1 : 2;
  const int ntiles = nk * world * NRB;
  for (int t = wave; t < ntiles; t += 16) {
    const int kk = t / (world * NRB);
    const int rem = t - kk * world * NRB;
    const int src = rem / NRB, jb = rem - src * NRB;
    const int kind = post ? 0 : (kk == 0 ? 2 : 1);
    const rsrc_t rgs = make_rsrc(pm_gg(m, pg, kind, src), PIPE_MAT * 4u);
    lds_byte_t* dst = (lds_byte_t*)(strips) + (size_t)t * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rgs, dst, 16, (int)(pipe_frag(rb, jb, lane) * 4u), 0, 0, 0);
  }
  // ---- the step counter, then every operand of the wave in one go: a tile wave asks for the boxes of step s-1
  //      right away (first two sources; they carry their own arrival marks), its output-layer wave for those of
  //      step s-2, whose correction it runs ----
  const rsrc_t rctr = make_rsrc(p.ctr, 16u);
  const int ls = bldi(rctr, (unsigned)(2 * a + 1) * 4u);
  const rsrc_t rq = make_rsrc(post ? p.Z1 + (size_t)pg * PIPE_MAT : p.Q + (size_t)a * PIPE_MAT, PIPE_MAT * 4u);
  float qv[4], db[2][NRB][4];
  bld4(rq, t_wave ? pipe_frag(rb, tile, lane) * 4u : kOOB, qv);
  if (t_wave) pm_load_d<NRB>(m, g1, 0, tile, true, lane, db);
  else pm_load_d<NRB>(m, g2, 0, tile, o_wave && !post, lane, db);
  const int srow = tid >> 4;                          // threads [0,256): row srow, class fr
  const int lslot = post ? ((p.q + PIPE_SLOTS - 1) & (PIPE_SLOTS - 1)) : p.q;
  const rsrc_t rlab = make_rsrc(p.labg + lslot * PIPE_LABG_SLOT, PIPE_LABG_SLOT * 4u);
  const rsrc_t rli = make_rsrc(p.labi + lslot * PIPE_ROWS, PIPE_ROWS * 4u);
  const float lab = bld(rlab, tid < 256 ? (unsigned)(fr * PIPE_ROWS + m0 + srow) * 4u : kOOB);
  const int lab_class = bldi(rli, tid < 256 ? (unsigned)(m0 + srow) * 4u : kOOB);
  const unsigned blk4 = (unsigned)p.HP * 64u;
  const rsrc_t rw2 = make_rsrc(p.W2m + (size_t)pg * p.HP * 16, blk4);
  float w2o[4];
  bld4(rw2, o_wave ? (unsigned)(tile * 64 + lane) * 16u : kOOB, w2o);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                     // the strips are in LDS
  PM_STAMP(1);
  if (!(wave < (NT > 4 ? NT : 4) || o_wave)) return;   // (waves 0..3 also carry the softmax rows)

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // ---- correction for D(s-2): nothing of this launch is needed; it runs on the tile's output-layer wave while the
  //      tile wave is already after the boxes of step s-1, and travels to the tile wave through LDS ----
  if (o_wave && !post) {
    pm_correct<NRB>(m, strips, 0, 0, lane, db, acc0, acc1);
    for (int s0 = 2; s0 < world; s0 += 2) {
      pm_load_d<NRB>(m, g2, s0, tile, true, lane, db);
      pm_correct<NRB>(m, strips, 0, s0, lane, db, acc0, acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acca[tile][r][lane] = acc0[r] + acc1[r];
  }
  PM_STAMP(2);
  // ---- the boxes of step s-1: what came back with the first request is looked at; words that still show the phase
  //      of four steps ago are asked for again ----
  PmPoll poll = {0ull, false, false};
  const int ph1 = pm_phase(ls - 1);
  if (t_wave) {
    // correction for D(s-1) (accuracy-only rows: the only one)
    const int kk = post ? 0 : 1;
    for (int s0 = 0; s0 < world; s0 += 2) {
      if (s0 > 0) pm_load_d<NRB>(m, g1, s0, tile, true, lane, db);
      pm_await_d<NRB>(m, g1, s0, tile, lane, ph1, poll, db);
      pm_correct<NRB>(m, strips, kk, s0, lane, db, acc0, acc1);
    }
  }
  PM_STAMP(3);
  float hreg[4] = {0.f, 0.f, 0.f, 0.f};
  float zv[4] = {0.f, 0.f, 0.f, 0.f};
  if (o_wave) {
    // U(s) rows n0 + 4g + j, class fr: U(s-1) minus lrN times every rank's, every row block's share, in rank order;
    // two sources per request
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < world; s0 += 2) {
      for (;;) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const bool on = s0 + u < world;
          const rsrc_t rp2 = make_rsrc(pm_p2(mine, g1, on ? s0 + u : 0, 0), 8u * PM_P2_BLK * 4u);
#pragma unroll
          for (int k = 0; k < NRB; ++k)
            pm_bld4_sys(rp2, on ? (unsigned)(tile * 64 + lane) * 16u : kOOB, (unsigned)k * PM_P2_BLK * 4u, db[u][k]);
        }
        const int bad = pm_first_stale<NRB>(m, s0, ph1, db);
        if (bad < 0) break;
        const int u = bad / NRB, k = bad - u * NRB;
        const rsrc_t rp2 = make_rsrc(pm_p2(mine, g1, s0 + u, 0), 8u * PM_P2_BLK * 4u);
        if (pm_wait_piece(m, poll, rp2, (unsigned)(tile * 64 + lane) * 16u, (unsigned)k * PM_P2_BLK * 4u, ph1, lane) == PM_GAVE_UP) break;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (s0 + u < world) {
#pragma unroll
          for (int k = 0; k < NRB; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[j] += db[u][k][j];
        }
    }
    float w2v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w2v[j] = w2o[j] - p.lr * sum[j];
      w2t[tile][4 * g + j][fr] = w2v[j];
    }
    if (rbl == 0) {                                    // one workgroup keeps U(s) for the next launch
      const f32x4 w4 = {w2v[0], w2v[1], w2v[2], w2v[3]};
      *reinterpret_cast<f32x4*>(p.W2m + (size_t)a * p.HP * 16 + (size_t)(tile * 64 + lane) * 4) = w4;
    }
  }
  PM_STAMP(4);
  __syncthreads();                                     // U(s) rows and the D(s-2) terms of every tile are in LDS
  if (t_wave) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      zv[r] = qv[r] - p.lr * ((post ? 0.f : acca[tile][r][lane]) + (acc0[r] + acc1[r]));
      // column Hd of h' is the constant 1 that carries b2; columns beyond it do not exist
      hreg[r] = col < p.Hd ? fmaxf(zv[r], 0.f) : (col == p.Hd ? 1.f : 0.f);
      hs[tile][4 * g + r][fr] = hreg[r];
    }
    if (!post) {                                       // z1 of this batch: the next launch's accuracy-only rows start from it
      const f32x4 z4 = {zv[0], zv[1], zv[2], zv[3]};
      *reinterpret_cast<f32x4*>(p.Z1 + (size_t)a * PIPE_MAT + pipe_frag(rb, tile, lane)) = z4;
    }
  }
  // ---- stage 2: partial logits of the tile's 16 hidden columns (the wave reads back only its own h tile: LDS
  //      operations of one wave execute in order, no workgroup barrier needed) ----
  float w2b[4] = {0.f, 0.f, 0.f, 0.f}, ha[4] = {0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_wave_barrier();
  if (t_wave) {
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
    // (MFMA step j takes hidden column n0 + 4g + j)
    z0 = MFMA16(hs[tile][fr][4 * g], w2t[tile][4 * g][fr], z0);
    z1 = MFMA16(hs[tile][fr][4 * g + 1], w2t[tile][4 * g + 1][fr], z1);
    z0 = MFMA16(hs[tile][fr][4 * g + 2], w2t[tile][4 * g + 2][fr], z0);
    z1 = MFMA16(hs[tile][fr][4 * g + 3], w2t[tile][4 * g + 3][fr], z1);
#pragma unroll
    for (int r = 0; r < 4; ++r) zp[tile][4 * g + r][fr] = z0[r] + z1[r];
    if (!post) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w2b[j] = w2t[tile][fr][4 * j + g];
        ha[j] = hs[tile][4 * j + g][fr];
      }
    }
  }
  __syncthreads();
  PM_STAMP(5);
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
    if (fr == 0) {
      row_loss[srow] = row_ok ? loss : 0.f;
      row_hit[srow] = row_ok ? hit : 0;
    }
  }
  __syncthreads();
  PM_STAMP(6);
  if (wave == 8) {
    // the row block's log entries, added up in row order by a wave that has nothing else left to do
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
  // ---- stage 4: dz1 tile = (dz2 W2^T) masked by the ReLU; stage 5: this row block's share of h'^T dz2 ----
  f32x4 d4, u4;
  {
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    d0 = MFMA16(dzs[fr][g], w2b[0], d0);
    d1 = MFMA16(dzs[fr][4 + g], w2b[1], d1);
    d0 = MFMA16(dzs[fr][8 + g], w2b[2], d0);
    d1 = MFMA16(dzs[fr][12 + g], w2b[3], d1);
#pragma unroll
    for (int r = 0; r < 4; ++r) d4[r] = (col < p.Hd && hreg[r] > 0.f) ? d0[r] + d1[r] : 0.f;
    f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
    u0 = MFMA16(ha[0], dzs[g][fr], u0);
    u1 = MFMA16(ha[1], dzs[4 + g][fr], u1);
    u0 = MFMA16(ha[2], dzs[8 + g][fr], u0);
    u1 = MFMA16(ha[3], dzs[12 + g][fr], u1);
#pragma unroll
    for (int r = 0; r < 4; ++r) u4[r] = u0[r] + u1[r];
  }
  // ---- the exchange: both results, every word marked with this step's phase, into the boxes of EVERY rank (the
  //      same bytes everywhere); nothing waits for the stores ----
  const int phw = pm_phase(ls);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d4[r] = pm_mark(d4[r], phw);
    u4[r] = pm_mark(u4[r], phw);
  }
#pragma unroll
  for (int d = 0; d < PM_MAXR; ++d) {
    if (d < world) {
      const rsrc_t rdd = make_rsrc(pm_d1(m.region[d], g0, m.rank), PIPE_MAT * 4u);
      const rsrc_t rpd = make_rsrc(pm_p2(m.region[d], g0, m.rank, rb), PM_P2_BLK * 4u);
      pm_store4_sys(rdd, pipe_frag(tile, rb, lane) * 4u, d4);
      pm_store4_sys(rpd, (unsigned)(tile * 64 + lane) * 16u, u4);
    }
  }
  PM_STAMP(7);
}

// ---------------------------------------------------------------------------------------------------------
// role Q: one 16x16 tile of Q(s+1) per 16-wave workgroup, split-K over the waves
// ---------------------------------------------------------------------------------------------------------
template <int NRB>
__device__ __forceinline__ void pm_role_q(const DtfPipeMultiArgs& m, int idx, float* smem) {
  const DtfPipeArgs& p = m.b;
  float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int NT = p.HP >> 4;
  const int rbq = idx / NT, nt = idx - rbq * NT;
  const bool post = rbq >= NRB;                        // prime launch only: z1 of the batch before the first step
  const int rb = post ? rbq - NRB : rbq;
  const int m0 = rb * 16, n0 = nt * 16;
  const int a = p.q & 1, pg = a ^ 1;
  const int g2 = (p.q + 2) & 3;
  const int slot = post ? p.q : p.q + 1;
  const unsigned slot_bytes = (unsigned)PIPE_ROWS * (unsigned)p.ldxa * 4u;
  const rsrc_t rx = make_rsrc(pm_xa(m, slot, m.rank), slot_bytes);
  const rsrc_t rw = make_rsrc(p.Wm + (size_t)pg * p.KAp * p.HP, (unsigned)p.KAp * (unsigned)p.HP * 4u);
  const int nX = p.KAp >> 4;
  const int nC = p.prime ? 0 : m.world * NRB;          // (prime: no pending gradient)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nX + nC; c0 += 64) {
    float av[4][4], bv[4][4], scale[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + wave + 16 * i;                // wave-uniform
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
      } else if (c - nX < nC) {
        const int cc = c - nX;
        const int src = cc / NRB, jb = cc - src * NRB;
        scale[i] = -p.lr;
        const rsrc_t rg = make_rsrc(pm_gg(m, pg, 3, src), PIPE_MAT * 4u);
        const rsrc_t rd = make_rsrc(pm_d1(m.region[m.rank], g2, src), PIPE_MAT * 4u);
        bld4(rg, pipe_frag(rb, jb, lane) * 4u, av[i]);
        pm_bld4_sys(rd, pipe_frag(nt, jb, lane) * 4u, 0, bv[i]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) av[i][j] = bv[i][j] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc0 = MFMA16(av[i][0] * scale[i], bv[i][0], acc0);
      acc1 = MFMA16(av[i][1] * scale[i], bv[i][1], acc1);
      acc0 = MFMA16(av[i][2] * scale[i], bv[i][2], acc0);
      acc1 = MFMA16(av[i][3] * scale[i], bv[i][3], acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
    const int e_rl = 4 * (e_ln >> 4) + e_r, e_cl = e_ln & 15;
    const float s = pipe_reduce16(red, e_r, e_ln);
    float* out = post ? p.Z1 + (size_t)a * PIPE_MAT : p.Q + (size_t)pg * PIPE_MAT;
    out[pipe_frag(rb, nt, (e_rl >> 2) * 16 + e_cl) + (e_rl & 3)] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// role G: two adjacent 16x16 tiles of a Gram matrix (rows: this rank's batch, columns: a source rank's batch) per
// 16-wave workgroup, split-K over the waves; the A operand (this rank's rows) is loaded once for both tiles
// ---------------------------------------------------------------------------------------------------------
template <int NRB>
__device__ __forceinline__ void pm_role_g(const DtfPipeMultiArgs& m, int idx, float* smem) {
  const DtfPipeArgs& p = m.b;
  float (*red)[16][4][64] = reinterpret_cast<float (*)[16][4][64]>(smem);      // [2 tiles][16 waves][4][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  constexpr int NP = (NRB + 1) / 2;                    // column pairs per row of tiles
  const int per = NRB * NP;
  const int ks = idx / per;                            // kind * world + source
  const int kind = ks / m.world, src = ks - kind * m.world;
  const int rem = idx - ks * per;
  const int rbi = rem / NP, cb0 = 2 * (rem - rbi * NP);
  const bool two = cb0 + 1 < NRB;                      // (odd NRB: the last pair is a single tile)
  const int m0 = rbi * 16, n0 = cb0 * 16;
  const int a = p.q & 1;
  // kinds: G(s,s), G(s+1,s), G(s+1,s-1), G(s+2,s-1)
  const int slot_a = p.q + (kind == 0 ? 0 : kind == 3 ? 2 : 1);
  const int slot_b = p.q + (kind < 2 ? 0 : PIPE_SLOTS - 1);
  const unsigned slot_bytes = (unsigned)PIPE_ROWS * (unsigned)p.ldxa * 4u;
  const rsrc_t ra = make_rsrc(pm_xa(m, slot_a, m.rank), slot_bytes);
  const rsrc_t rb = make_rsrc(pm_xa(m, slot_b, src), slot_bytes);
  const int nX = p.KAp >> 4;
  float av[4][4], bv[2][4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = wave + 16 * i;
    const unsigned k4 = (unsigned)(16 * c + 4 * g) * 4u;
    bld4(ra, c < nX ? (unsigned)(m0 + fr) * (unsigned)p.ldxa * 4u + k4 : kOOB, av[i]);
    bld4(rb, c < nX ? (unsigned)(n0 + fr) * (unsigned)p.ldxa * 4u + k4 : kOOB, bv[0][i]);
    bld4(rb, (c < nX && two) ? (unsigned)(n0 + 16 + fr) * (unsigned)p.ldxa * 4u + k4 : kOOB, bv[1][i]);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc0 = MFMA16(av[i][0], bv[t][i][0], acc0);
      acc1 = MFMA16(av[i][1], bv[t][i][1], acc1);
      acc0 = MFMA16(av[i][2], bv[t][i][2], acc0);
      acc1 = MFMA16(av[i][3], bv[t][i][3], acc1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[t][wave][r][lane] = acc0[r] + acc1[r];
  }
  __syncthreads();
  if (threadIdx.x < 512) {
    const int t = threadIdx.x >> 8;
    if (t == 1 && !two) return;
    const int e_r = (threadIdx.x >> 6) & 3, e_ln = threadIdx.x & 63;
    const int e_rl = 4 * (e_ln >> 4) + e_r, e_cl = e_ln & 15;
    const float s = pipe_reduce16(red[t], e_r, e_ln);
    float* out = pm_gg(m, a, kind, src);
    out[pipe_frag(rbi, cb0 + t, (e_cl >> 2) * 16 + e_rl) + (e_cl & 3)] = s;   // A-operand layout
  }
}

// ---------------------------------------------------------------------------------------------------------
// role W: Wm[a] = Wm[p] - lrN sum over source ranks of X'_src(s-2)^T D_src(s-2); a tile is split over `ks` waves by
// source rank (source = wave's share, + ks, ...), partial sums through LDS in a fixed order.  Flush: both pending
// updates (steps q-2 and q-1 of the slot after the run's last step), result into W1 / b1 of the flat buffer.
// ---------------------------------------------------------------------------------------------------------
template <int NRB>
__device__ __forceinline__ void pm_role_w(const DtfPipeMultiArgs& m, int idx, float* smem, int ksplit) {
  const DtfPipeArgs& p = m.b;
  float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int NT = p.HP >> 4, KT = p.KAp >> 4;
  const int tpw = 16 / ksplit;
  const int tl = wave / ksplit, kshare = wave - tl * ksplit;
  const int tile = idx * tpw + tl;
  const bool valid = tile < KT * NT;                   // wave-uniform
  const int mt = valid ? tile / NT : 0, nt = valid ? tile - mt * NT : 0;
  const int m0 = mt * 16, n0 = nt * 16;
  const int a = p.q & 1, pg = a ^ 1;
  const unsigned slot_bytes = (unsigned)PIPE_ROWS * (unsigned)p.ldxa * 4u;
  const rsrc_t rw = make_rsrc(p.Wm + (size_t)pg * p.KAp * p.HP, (unsigned)p.KAp * (unsigned)p.HP * 4u);
  const unsigned ldx4 = (unsigned)p.ldxa * 4u;
  const unsigned a_base = (unsigned)(4 * g) * ldx4 + (unsigned)(m0 + fr) * 4u;
  float wold[4];
  bld4(rw, (valid && kshare == 0) ? pipe_wfrag(mt, nt, NT, lane) * 4u : kOOB, wold);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nterm = p.flush ? 2 : 1;
  for (int t = 0; t < nterm; ++t) {
    const int step = p.q + PIPE_SLOTS - 2 + t;         // slot of step s-2 (flush: then s-1 of the slot after the last step)
    for (int src = kshare; src < m.world; src += ksplit) {
      const rsrc_t rx = make_rsrc(pm_xa(m, step, src), slot_bytes);
      const rsrc_t rd = make_rsrc(pm_d1(m.region[m.rank], step & 3, src), PIPE_MAT * 4u);
      float av[NRB][4], bv[NRB][4];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        pm_bld4_sys(rd, valid ? pipe_frag(nt, rb, lane) * 4u : kOOB, 0, bv[rb]);
#pragma unroll
        for (int j = 0; j < 4; ++j) av[rb][j] = pipe_bld_s(rx, valid ? a_base : kOOB, (unsigned)(16 * rb + j) * ldx4);
      }
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        acc0 = MFMA16(av[rb][0], bv[rb][0], acc0);
        acc1 = MFMA16(av[rb][1], bv[rb][1], acc1);
        acc0 = MFMA16(av[rb][2], bv[rb][2], acc0);
        acc1 = MFMA16(av[rb][3], bv[rb][3], acc1);
      }
    }
  }
  float sum[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = acc0[r] + acc1[r];
  if (ksplit > 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = sum[r];
    __syncthreads();
    if (kshare != 0) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = 0.f;
      for (int k = 0; k < ksplit; ++k) s += red[tl * ksplit + k][r][lane];     // fixed order: the same bits on every rank
      sum[r] = s;
    }
  }
  if (!valid) return;
  const int col = n0 + fr;
  f32x4 w4;
#pragma unroll
  for (int r = 0; r < 4; ++r) w4[r] = wold[r] - p.lr * sum[r];
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
// role T: counters, log header, gather of EVERY rank's batch of step s+3, row indices of step s+4
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pm_role_t(const DtfPipeMultiArgs& m, int tb) {
  const DtfPipeArgs& p = m.b;
  const DtfTailArgs& t = p.tail;
  const int tid = threadIdx.x;
  const int a = p.q & 1;
  const int B = p.B, C = p.C;
  const int per = (B + 7) >> 3;
  const int src = tb / per;                            // whose batch this workgroup gathers
  const int sub = tid >> 7, l = tid & 127;
  const int i = 8 * (tb - src * per) + sub;
  const int q3 = p.q + 3, q4 = p.q + 4;
  const bool own = src == m.rank;
  const int row = i < B ? pm_ridx(m, q3, src)[i] : 0;
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
  const f32x4* srcp = reinterpret_cast<const f32x4*>(p.X + (size_t)row * p.ldx);
  float* drow = const_cast<float*>(pm_xa(m, q3, src)) + (size_t)i * p.ldxa;
  f32x4* dst = reinterpret_cast<f32x4*>(drow);
  const int n4 = p.In >> 2;
  for (int base = 0; base < n4; base += 128 * 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = base + 128 * u + l;
      v[u] = srcp[k < n4 ? k : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = base + 128 * u + l;
      if (k < n4) dst[k] = v[u];
    }
  }
  if (l == 17) drow[p.In] = 1.f;                       // the column that carries the bias row of W1'
  if (own && l < 16) {
    const int s3 = q3 & (PIPE_SLOTS - 1);
    const float v = l < C ? p.labels[(size_t)row * C + l] : -__builtin_huge_valf();
    p.labg[s3 * PIPE_LABG_SLOT + l * PIPE_ROWS + i] = l < C ? v : 0.f;
    // class of the row for the accuracy count: argmax with lowest-index tie break (tf.argmax, distributed.py:83)
    const float vmax = row16_max(v);
    const int am = row16_min(v == vmax ? l : 16);
    if (l == 0) p.labi[s3 * PIPE_ROWS + i] = am;
  }
  if (l == 16) {
    const long long pos64 = (long long)(ls + 4) * B + i;
    const long long epoch = pos64 / t.n_train;
    const int pos = (int)(pos64 - epoch * t.n_train);
    int* rn = pm_ridx(m, q4, src);
    const int* perm = m.perm[src];
    rn[i] = perm ? perm[(size_t)(epoch & 1) * t.n_train + pos] : pos;
    rn[B + i] = row;                                   // the batch before it: its accuracy-only rows (host use)
  }
}

// role F (flush): W2, b2 of the flat buffer = W2m[p] - lrN sum over ranks and row blocks of the last step's boxes
template <int NRB>
__device__ __forceinline__ void pm_role_f(const DtfPipeMultiArgs& m) {
  const DtfPipeArgs& p = m.b;
  const int pg = (p.q & 1) ^ 1;
  const int g1 = (p.q + 3) & 3;
  for (int e = threadIdx.x; e < p.HP * 16; e += 1024) {
    float s = 0.f;
    for (int src = 0; src < m.world; ++src) {
      const float* box = pm_p2(m.region[m.rank], g1, src, 0);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) s += __hip_atomic_load(box + (size_t)rb * PM_P2_BLK + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float w = p.W2m[(size_t)pg * p.HP * 16 + e] - p.lr * s;
    const int ln = (e >> 2) & 63;
    const int k = (e >> 8) * 16 + (ln >> 4) * 4 + (e & 3), c = ln & 15;      // inverse of pipe_w2_off
    if (c < p.C) {
      if (k < p.Hd) p.W2[(size_t)k * p.C + c] = w;
      else if (k == p.Hd) p.b2[c] = w;
    }
  }
}

// Role C (prime launch): both generations of the working copies from the flat parameter buffer, this rank's boxes
// zero (no pending update from anybody), its flags at the number of steps done, counters of the first step.
__device__ __forceinline__ void pm_role_c(const DtfPipeMultiArgs& m, int cb) {
  const DtfPipeArgs& p = m.b;
  const long long nWm = (long long)p.KAp * p.HP, nW2 = (long long)p.HP * 16;
  const long long nBoxD = (long long)PM_GENS * PIPE_MAT, nBoxP = (long long)PM_GENS * 8 * PM_P2_BLK;
  const long long total = pm_copy_elems(m);
  char* const mine = m.region[m.rank];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    long long e = (long long)cb * 4096 + u * 1024 + (int)threadIdx.x;
    if (e >= total) break;
    if (e < 2 * nWm) {
      const int w = (int)(e >= nWm ? e - nWm : e);
      const int NTc = p.HP >> 4;
      const int tl = w >> 8, ln = (w >> 2) & 63;
      const int k = (tl / NTc) * 16 + (ln >> 4) * 4 + (w & 3), c = (tl % NTc) * 16 + (ln & 15);    // inverse of pipe_wfrag
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
      const int w = (int)(e >= nW2 ? e - nW2 : e);
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
    // boxes of source rank `src`: every generation of D, then every generation of the output-layer shares
    const int src = (int)(e / (nBoxD + nBoxP));
    const long long r = e - (long long)src * (nBoxD + nBoxP);
    // zero, marked: the generation the first step reads as "the step before it" has arrived (no pending update from
    // anybody), every other generation shows the opposite of the phase its next writer will bring
    const int first = p.tail.state[1];
    if (r < nBoxD) {
      const int gen = (int)(r / PIPE_MAT);
      pm_d1(mine, gen, src)[r - (long long)gen * PIPE_MAT] = pm_prime_word(first, gen);
    } else {
      const long long rr = r - nBoxD;
      const int gen = (int)(rr / (8 * PM_P2_BLK));
      pm_p2(mine, gen, src, 0)[rr - (long long)gen * 8 * PM_P2_BLK] = pm_prime_word(first, gen);
    }
  }
  if (cb == 0 && threadIdx.x < 2) {
    // p.q is the slot of the launch BEFORE the first step: that step reads generation (q + 1) & 1 of the counters
    p.ctr[2 * ((p.q & 1) ^ 1) + threadIdx.x] = p.tail.state[threadIdx.x];
  }
}

template <int NRB>
__global__ __launch_bounds__(1024) void pipem_step_kernel(DtfPipeMultiArgs m) {
  extern __shared__ __attribute__((aligned(16))) float pm_smem[];
  const PmDims d = pm_dims(m, NRB);
  // role index XCD-major with the roles in the order L G Q W T while the grid is resident at once, as in
  // pipe_step_kernel (mlp_pipe.hip), where it is measured; plain_order: several ranks rehearse on ONE GPU, where the
  // row workgroups of all of them would crowd XCD 0
  const bool xcd_major = gridDim.x <= 512 && !m.plain_order;
  int b = xcd_major ? dtf_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  if (b < d.nL) { pm_role_l<NRB>(m, b, pm_smem); return; }
  b -= d.nL;
  if (xcd_major) {
    if (b < d.nG) { pm_role_g<NRB>(m, b, pm_smem); return; }
    b -= d.nG;
    if (b < d.nQ) { pm_role_q<NRB>(m, b, pm_smem); return; }
    b -= d.nQ;
    if (b < d.nW) { pm_role_w<NRB>(m, b, pm_smem, d.ks); return; }
    b -= d.nW;
    if (b < d.nT) { pm_role_t(m, b); return; }
    b -= d.nT;
  } else {
    if (b < d.nT) { pm_role_t(m, b); return; }
    b -= d.nT;
    if (b < d.nW) { pm_role_w<NRB>(m, b, pm_smem, d.ks); return; }
    b -= d.nW;
    if (b < d.nQ) { pm_role_q<NRB>(m, b, pm_smem); return; }
    b -= d.nQ;
    if (b < d.nG) { pm_role_g<NRB>(m, b, pm_smem); return; }
    b -= d.nG;
  }
  if (b < d.nF) { pm_role_f<NRB>(m); return; }
  b -= d.nF;
  if (b < d.nC) pm_role_c(m, b);
}

// One workgroup in front of the flush launch: returns when every rank's boxes of the last step have arrived (every
// word of them shows that step's phase).  A wave looks at four 1 KB pieces per request.
__global__ __launch_bounds__(1024) void pipem_wait_kernel(DtfPipeMultiArgs m, int nrb) {
  const DtfPipeArgs& p = m.b;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NT = p.HP >> 4;
  const int last = p.tail.state[1] - 1;               // the run's last step
  const int phase = pm_phase(last), gen = last & 3;
  char* const mine = m.region[m.rank];
  const int per_src = 2 * nrb * NT;                    // D tiles, then output-layer tiles
  const int total = m.world * per_src;
  PmPoll st = {0ull, false, false};
  for (int f0 = wave * 4; f0 < total; f0 += 64) {
    for (;;) {
      unsigned stale = 0;
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int f = f0 + u;
        const bool on = f < total;
        const int src = on ? f / per_src : 0, r = on ? f - src * per_src : 0;
        const bool is_d = r < nrb * NT;
        const int rr = is_d ? r : r - nrb * NT;
        const int tile = rr / nrb, rb = rr - tile * nrb;
        const float* base = is_d ? pm_d1(mine, gen, src) + pipe_frag(tile, rb, 0) : pm_p2(mine, gen, src, rb) + (size_t)tile * 256;
        const rsrc_t rs = make_rsrc(base, 1024u);
        pm_bld4_sys(rs, on ? (unsigned)lane * 16u : kOOB, 0, v[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (f0 + u < total) stale |= pm_stale4(v[u], phase);
      if (pm_poll(m, st, stale, lane) != PM_AGAIN) break;
    }
  }
}

static int pm_nrb_template(int B) {
  const int nrb = dtf_cdiv(B, 16);
  return nrb <= 2 ? 2 : nrb <= 4 ? 4 : nrb <= 7 ? 7 : 8;
}
static size_t pm_lds_bytes(const DtfPipeMultiArgs& a) {
  return (size_t)PM_SMEM_BASE * 4 + (size_t)2 * a.world * pm_nrb_template(a.b.B) * 1024;
}

static int pm_validate(const DtfPipeMultiArgs* m) {
  if (!m) return DTF_ERR_ARG;
  const DtfPipeArgs* a = &m->b;
  if (!a->X || !a->labels || !a->Xa || !a->labg || !a->ridx || !a->Wm || !a->W2m || !a->Q || !a->GG || !a->ctr || !a->Z1 ||
      !a->labi || !a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->wlog || !a->tail.state || !m->err)
    return DTF_ERR_ARG;
  if (m->world < 1 || m->world > PM_MAXR || m->rank < 0 || m->rank >= m->world) return DTF_ERR_ARG;
  for (int r = 0; r < m->world; ++r)
    if (!m->region[r] || ((uintptr_t)m->region[r] & 15)) return DTF_ERR_ARG;
  if (a->In <= 0 || (a->In & 3) || a->Hd <= 0 || a->Hd > 127 || a->C <= 0 || a->C > 16 || a->B <= 0 || a->B > PIPE_ROWS || a->td != 8)
    return DTF_ERR_SHAPE;
  if (a->R != a->B && a->R != 2 * a->B) return DTF_ERR_SHAPE;
  if (a->HP != ((a->Hd + 1 + 15) & ~15) || a->KAp != ((a->In + 1 + 15) & ~15) || a->ldxa != a->KAp) return DTF_ERR_SHAPE;
  if ((a->KAp >> 4) > 64) return DTF_ERR_SHAPE;                                 // role G: 16 waves x 4 chunks of X
  if (a->q < 0 || a->q >= PIPE_SLOTS || a->ldx < a->In || (a->ldx & 3) || a->log_cap <= 0 || a->ridx_len < 2 * a->B)
    return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->X) || !dtf_aligned16(a->Xa) || !dtf_aligned16(a->GG) || !dtf_aligned16(a->Q) || !dtf_aligned16(a->Z1) ||
      !dtf_aligned16(a->Wm) || !dtf_aligned16(a->W2m))
    return DTF_ERR_ALIGN;
  if ((uint64_t)PIPE_ROWS * a->ldxa * 4u >= 0xFFFFFFF0u || (uint64_t)a->KAp * a->HP * 4u >= 0xFFFFFFF0u) return DTF_ERR_SHAPE;
  if (a->tail.n_train <= 0 || !a->tail.enable) return DTF_ERR_SHAPE;
  if (a->roles <= 0 || a->roles > 127) return DTF_ERR_ARG;
  if (a->flush && (a->roles & ~(PIPE_ROLE_W | PIPE_ROLE_F))) return DTF_ERR_ARG;
  if (a->prime && (a->flush || a->roles != (PIPE_ROLE_Q | PIPE_ROLE_C))) return DTF_ERR_ARG;
  if ((a->roles & PIPE_ROLE_C) && !a->prime) return DTF_ERR_ARG;
  if (pm_lds_bytes(*m) > 160u * 1024u) return DTF_ERR_SHAPE;                    // Gram strips of one row block: LDS of one CU
  return DTF_OK;
}

template <int NRB>
static int pm_launch(const DtfPipeMultiArgs& a, hipStream_t s) {
  const PmDims d = pm_dims(a, NRB);
  const int nblk = d.nL + d.nQ + d.nG + d.nW + d.nT + d.nF + d.nC;
  const size_t lds = pm_lds_bytes(a);
  static size_t allowed = 0;                           // per instantiation: the largest dynamic LDS size asked for so far
  if (lds > allowed) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pipem_step_kernel<NRB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return DTF_ERR_LAUNCH;
    }
    allowed = lds;
  }
  hipLaunchKernelGGL((pipem_step_kernel<NRB>), dim3(nblk), dim3(1024), lds, s, a);
  return dtf_check_launch();
}

static int pm_launch_any(const DtfPipeMultiArgs& a, hipStream_t s) {
  switch (pm_nrb_template(a.b.B)) {
    case 2: return pm_launch<2>(a, s);
    case 4: return pm_launch<4>(a, s);
    case 7: return pm_launch<7>(a, s);
    default: return pm_launch<8>(a, s);
  }
}

DTF_EXPORT long long dtf_pipem_region_bytes(void) { return (long long)DTF_PIPEM_REGION_BYTES; }

// One step: every role.  `a->b.q` = slot of the step.
DTF_EXPORT int dtf_pipem_step(const DtfPipeMultiArgs* a, hipStream_t s) {
  const int rc = pm_validate(a);
  if (rc != DTF_OK) return rc;
  if (a->b.flush || a->b.prime || a->b.roles != (PIPE_ROLE_L | PIPE_ROLE_Q | PIPE_ROLE_G | PIPE_ROLE_W | PIPE_ROLE_T)) return DTF_ERR_ARG;
  return pm_launch_any(*a, s);
}

// Start of a run whose first step has slot (q + 1) & 7.  The caller keeps every rank from starting that step until
// every rank's prime launch has finished (a rank's boxes are zeroed here; a peer's first step writes into them).
DTF_EXPORT int dtf_pipem_prime(const DtfPipeMultiArgs* a, hipStream_t s) {
  const int rc = pm_validate(a);
  if (rc != DTF_OK) return rc;
  if (a->b.flush || !a->b.prime || a->b.roles != (PIPE_ROLE_Q | PIPE_ROLE_C)) return DTF_ERR_ARG;
  return pm_launch_any(*a, s);
}

// End of a run whose last step had slot (q - 1) & 7: a one-wave launch that waits for the last step's boxes of every
// rank, then the parameters after that step into the flat buffer.
DTF_EXPORT int dtf_pipem_flush(const DtfPipeMultiArgs* a, hipStream_t s) {
  const int rc = pm_validate(a);
  if (rc != DTF_OK) return rc;
  if (!a->b.flush || a->b.prime || a->b.roles != (PIPE_ROLE_W | PIPE_ROLE_F)) return DTF_ERR_ARG;
  hipLaunchKernelGGL(pipem_wait_kernel, dim3(1), dim3(1024), 0, s, *a, pm_nrb_template(a->b.B));
  const int rc2 = dtf_check_launch();
  if (rc2 != DTF_OK) return rc2;
  return pm_launch_any(*a, s);
}

// Measurement only: any subset of the step roles in one launch, results meaningless.
DTF_EXPORT int dtf_pipem_launch_roles(const DtfPipeMultiArgs* a, hipStream_t s) {
  const int rc = pm_validate(a);
  if (rc != DTF_OK) return rc;
  return pm_launch_any(*a, s);
}
