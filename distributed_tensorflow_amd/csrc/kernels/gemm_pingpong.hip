// bf16 GEMM  C[M,N] = A[M,K] . B[N,K]^T  for the large aligned products of the wide layers
// (M, N, K multiples of 8; it pays from ~192 blocks of 256x256 on): 256x256 block, 8 waves in two groups that alternate on the
// SIMDs ("ping-pong"): while one group issues its MFMAs, the other issues fragment reads and staging loads.
//
// Layout of the work
//   * 8 waves = 2 (rows) x 4 (columns); a wave owns a 128x64 block of C = 8x4 MFMA tiles of 16x16, held as four
//     quadrants (ha, hb) of 64x32.  Waves w and w+4 (the two row groups) share a SIMD.
//   * A K-tile is 64 deep and lives in LDS as four half-tiles of 128 rows x 128 bytes: A0/A1 hold, for both
//     wave rows, the first/second 64 rows of the wave's 128; B0/B1 hold, for all four wave columns, the
//     first/second 32 columns of the wave's 64.  Two K-tile buffers: 2 x 4 x 16 KB = 128 KB, one array.
//   * Staging is direct to LDS (buffer_load ... lds, 16 bytes per lane, no VGPR destination).  One
//     wave-instruction fills 8 consecutive 128-byte rows; the LDS image is lane-linear, so the XOR swizzle
//     (16-byte slot ^ (row & 7), which makes the ds_read_b128 fragment reads conflict-free) is applied on the
//     SOURCE address: the lane that fills physical slot s of row r fetches k-chunk s ^ (r & 7).
//
// Schedule: a K-tile takes 4 phases, one quadrant (16 MFMAs = 64x32 x K 64) each, in the order
//   (0,0) (0,1) (1,1) (1,0), so that a phase reads at most one new operand half:
//     phase 1: read B0 (4 x ds_read_b128) and A0 (8)     phase 2: read B1 (4)
//     phase 3: read A1 (8)                               phase 4: nothing (B0 is kept in registers)
//   and every phase stages ONE half-tile (2 loads per lane) of a later K-tile:
//     phases 1..8 of an iteration (K-tiles E = 2it in buffer 0, O = 2it+1 in buffer 1) stage
//       B1(O) A1(O) A0(E+2) B0(E+2) B1(E+2) A1(E+2) A0(O+2) B0(O+2).
//   Each phase is   reads ; stage ; s_waitcnt vmcnt(8) ; barrier ; 16 MFMAs ; barrier.
//   (Raising the wave priority around the MFMAs, as such schedules often do, measured 1-3 % slower here.)
//   The second wave group executes one barrier more at the start, so it runs half a phase behind the first:
//   its MFMAs coincide with the first group's reads/staging and vice versa.
//
// Why the schedule is race-free (phases counted continuously, p):
//   * vmcnt(8) after the stage of phase p leaves the stages of phases p-3..p outstanding, i.e. the half-tile
//     staged in phase s has landed, for the waiting wave, in phase s+4; it is read in phase s+5 or later.  Between
//     that wait and any read lies a barrier that every wave of BOTH groups has passed after its own wait (the late
//     group waits half a phase later, still before the first group's reads of the next phase).
//   * a half-tile region is overwritten two or more phases after the last phase that read it (A0: read 1,
//     restaged 3; B0: 1 -> 4; B1: 2 -> 5; A1: 3 -> 6, same +4 for the odd buffer), so the late group's reads, issued
//     half a phase after the early group's, have returned (the compiler waits for them before the MFMAs that follow
//     the next barrier) before any wave issues the overwriting load.
//   * stages past the last K-tile (and chunks past K inside it) are issued with an out-of-range offset (the buffer
//     range check delivers zeros), which keeps the counting uniform; vmcnt(0) before the epilogue: no load may land after the workgroup has ended.
#include <cstdlib>

#include "dtf_kernels.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
static constexpr unsigned kOOB = 0xFFFFFFF0u;

enum { EPI_FWD = 0, EPI_BWD = 1, EPI_WGRAD = 2, EPI_WGRAD_OPT = 3 };   // 3: weight gradient with the optimizer folded in
enum { REG_A0 = 0, REG_A1 = 1, REG_B0 = 2, REG_B1 = 3 };

__device__ __forceinline__ unsigned short pp_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float pp_bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// Diagnostic build only (-DDTF_STAMPS): DTF_GEMM_SKIP_STORES = 1 / 2 / 3 leaves out the global stores of C / of the
// transposed copy / of both in the forward and backward epilogues (everything else of the epilogue still runs): how
// much of a launch the final burst of stores costs, i.e. the most that overlapping it with another tile's main loop
// could win (tools/gemm_store_bound.sh).
#ifdef DTF_STAMPS
#define PP_DIAG_KEEP(bit) && !(p.pad_ & (bit))
#else
#define PP_DIAG_KEEP(bit)
#endif

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pp_kernel(DtfGemmArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 4 * 16384];   // [buffer][A0 A1 B0 B1][128 rows x 128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, g = lane >> 4;

  // M and N need not be multiples of 256: operand rows past the end are read out of range (zeros) or belong to a
  // larger buffer (finite or not, they only feed output rows/columns that are never stored), the epilogue guards
  const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + 255) >> 8;
  // split-K (plain weight gradient only): workgroups [s * tiles, (s + 1) * tiles) take K range s
  const int ntile = tiles_m * tiles_n;
  const int split = (EPI == EPI_WGRAD && p.ksplit > 1) ? (int)blockIdx.x / ntile : 0;
  const int kper = (EPI == EPI_WGRAD && p.ksplit > 1) ? (((p.K + p.ksplit - 1) / p.ksplit + 127) & ~127) : p.K;
  const int k_begin = split * kper;
  const int k_len = min(p.K - k_begin, kper);          // <= 0 for a split past the end: every load is out of range
  const int wg = dtf_xcd_remap((int)blockIdx.x - split * ntile, ntile);
  // consecutive (remapped) ids walk down a band of 8 row-tiles: the A panels and one B panel stay in that XCD's L2
  const int band = 8;
  const int per_band = band * tiles_n;
  const int b_idx = wg / per_band, in_band = wg % per_band;
  const int rows_in_band = min(band, tiles_m - b_idx * band);
  const int tm = b_idx * band + in_band % rows_in_band;
  const int tn = in_band / rows_in_band;
  const int m0 = tm << 8, n0 = tn << 8;

  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)p.b_bytes, 0x00020000);

  // ---- staging sources.  Instruction j (0,1) of this wave fills local rows (8j + wave)*8 .. +7 of a half-tile.
  //   A half h: local row lr <-> block row (lr >> 6)*128 + h*64 + (lr & 63);  here lr >> 6 == j, lr & 63 == wave*8 + srow
  //   B half h: local row lr <-> block col (lr >> 5)*64 + h*32 + (lr & 31);   here lr >> 5 == 2j + (wave >> 2)
  const int srow = lane >> 3;
  const unsigned kchunk = (unsigned)((lane & 7) ^ srow) * 16u;
  unsigned a_vo[2], b_vo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    a_vo[j] = (unsigned)(m0 + j * 128 + wave * 8 + srow) * (unsigned)p.lda * 2u + kchunk + (unsigned)k_begin * 2u;
    b_vo[j] = (unsigned)(n0 + (2 * j + (wave >> 2)) * 64 + (wave & 3) * 8 + srow) * (unsigned)p.ldb * 2u + kchunk + (unsigned)k_begin * 2u;
  }
  const unsigned a_half = 64u * (unsigned)p.lda * 2u, b_half = 32u * (unsigned)p.ldb * 2u;
  // K-tiles: an even number that covers K; 16-byte chunks at or past K are requested out of range (the range check
  // of the buffer load then delivers zeros), so K only has to be a multiple of 8
  const int nk = ((max(k_len, 0) + 127) >> 7) << 1;
  const int klim = k_len * 2 - (int)kchunk;          // a lane's chunk of K-tile kt is inside its K range iff kt*128 < klim
  lds_byte_t* const lds3 = (lds_byte_t*)(lds) + wave * 1024;

  auto stage = [&](int buf, int region, int kt) {
    const bool live = kt * 128 < klim;               // also false for every K-tile past the last
    const unsigned half = (unsigned)(region & 1);
    const unsigned so = (unsigned)kt * 128u + half * (region < REG_B0 ? a_half : b_half);
    lds_byte_t* dst = lds3 + buf * 65536 + region * 16384;
    if (region < REG_B0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, (int)(live ? a_vo[0] : kOOB), (int)so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst + 8192, 16, (int)(live ? a_vo[1] : kOOB), (int)so, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, (int)(live ? b_vo[0] : kOOB), (int)so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst + 8192, 16, (int)(live ? b_vo[1] : kOOB), (int)so, 0, 0);
    }
  };

  // ---- fragment reads: A tile i of half ha: local row wr*64 + 16i + fr; B tile jn of half hb: wc*32 + 16jn + fr;
  //      k-step ks: logical slot 4ks + g, physical slot ^ (fr & 7); the two k-steps differ by byte-offset bit 6
  const unsigned sw = (unsigned)(fr & 7);
  const unsigned a_l0 = (unsigned)(wr * 64 + fr) * 128u + (((unsigned)g ^ sw) << 4);
  const unsigned b_l0 = (unsigned)(wc * 32 + fr) * 128u + (((unsigned)g ^ sw) << 4);
  const unsigned a_l1 = a_l0 ^ 64u, b_l1 = b_l0 ^ 64u;

  f32x4 acc[2][4][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bias_grad[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const bool want_bias = (EPI == EPI_WGRAD || EPI == EPI_WGRAD_OPT) && p.bias_out != nullptr && tm == 0 && wr == 0;   // wave-uniform

  u32x4 af[4][2], bq0[2][2], bq1[2][2];

#define PP_READ_A(BUF, HA)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
    af[i][0] = *reinterpret_cast<const u32x4*>(&lds[(BUF) * 65536 + (HA) * 16384 + i * 2048 + a_l0]);   \
    af[i][1] = *reinterpret_cast<const u32x4*>(&lds[(BUF) * 65536 + (HA) * 16384 + i * 2048 + a_l1]);   \
  }
#define PP_READ_B(BUF, HB, BQ)                                                                          \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
    BQ[j][0] = *reinterpret_cast<const u32x4*>(&lds[(BUF) * 65536 + (HB) * 16384 + j * 2048 + b_l0]);   \
    BQ[j][1] = *reinterpret_cast<const u32x4*>(&lds[(BUF) * 65536 + (HB) * 16384 + j * 2048 + b_l1]);   \
  }
#define PP_BIAS(HB, BQ)                                                                                 \
  if (want_bias) {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                       \
      bias_grad[HB][j] += pp_bf2f((unsigned short)(BQ[j][ks][e] & 0xffff)) + pp_bf2f((unsigned short)(BQ[j][ks][e] >> 16)); \
  }
#define PP_MMA(HA, HB, BQ)                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
    acc[HA][i][HB][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i][ks]), \
                                                                 __builtin_bit_cast(bf16x8_t, BQ[j][ks]), acc[HA][i][HB][j], 0, 0, 0);
// one phase: READS ; stage ; counted wait ; barrier ; MFMAs ; barrier
#define PP_PHASE(READS, ST_BUF, ST_REG, ST_KT, HA, HB, BQ, BIAS)                                        \
  READS                                                                                                 \
  stage(ST_BUF, ST_REG, ST_KT);                                                                         \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_barrier();                                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  PP_MMA(HA, HB, BQ)                                                                                    \
  BIAS                                                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_barrier();                                                                         \
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: what the steady state has in flight at the start of an iteration
  stage(0, REG_A0, 0); stage(0, REG_B0, 0); stage(0, REG_B1, 0); stage(0, REG_A1, 0);
  stage(1, REG_A0, 1); stage(1, REG_B0, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // this group now runs half a phase behind the other
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = 0; kt < nk; kt += 2) {
    PP_PHASE(PP_READ_B(0, REG_B0, bq0) __builtin_amdgcn_sched_barrier(0); PP_READ_A(0, REG_A0),
             1, REG_B1, kt + 1, 0, 0, bq0, PP_BIAS(0, bq0))
    PP_PHASE(PP_READ_B(0, REG_B1, bq1), 1, REG_A1, kt + 1, 0, 1, bq1, PP_BIAS(1, bq1))
    PP_PHASE(PP_READ_A(0, REG_A1), 0, REG_A0, kt + 2, 1, 1, bq1, )
    PP_PHASE(, 0, REG_B0, kt + 2, 1, 0, bq0, )
    PP_PHASE(PP_READ_B(1, REG_B0, bq0) __builtin_amdgcn_sched_barrier(0); PP_READ_A(1, REG_A0),
             0, REG_B1, kt + 2, 0, 0, bq0, PP_BIAS(0, bq0))
    PP_PHASE(PP_READ_B(1, REG_B1, bq1), 0, REG_A1, kt + 2, 0, 1, bq1, PP_BIAS(1, bq1))
    PP_PHASE(PP_READ_A(1, REG_A1), 1, REG_A0, kt + 3, 1, 1, bq1, )
    PP_PHASE(, 1, REG_B0, kt + 3, 1, 0, bq0, )
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may land in LDS after this workgroup has ended
  if (wr == 0) __builtin_amdgcn_s_barrier();          // same number of barriers in both groups
#undef PP_PHASE
#undef PP_MMA
#undef PP_BIAS
#undef PP_READ_A
#undef PP_READ_B

  // ---- epilogue: acc[ha][i][hb][j][r] = C[m0 + wr*128 + ha*64 + 16i + 4g + r][n0 + wc*64 + hb*32 + 16j + fr]
  // After the last barrier no wave reads operands any more, and every wave turns its own 16 KB of LDS into a
  // transposition buffer for its 128x64 block: the MFMA result layout (a lane holds 4 rows of one column) would
  // otherwise store 2-byte pieces; from LDS every global store is 16 bytes per lane and whole 128/256-byte runs.
  typedef unsigned short T;
  unsigned char* const wl = lds + wave * 16384;
  const int crow = m0 + wr * 128, ccol = n0 + wc * 64;          // origin of this wave's block
  if (EPI == EPI_WGRAD || EPI == EPI_WGRAD_OPT) {
    float* __restrict__ Cf = (EPI == EPI_WGRAD && p.ksplit > 1) ? p.split_ws + (size_t)split * p.M * p.ldc : static_cast<float*>(p.C);
    // optimizer folded into the epilogue: the same arithmetic as optim.hip (adam_one / sgd), element for element
    constexpr bool opt = EPI == EPI_WGRAD_OPT;
    const bool adam = p.opt_m != nullptr;
    float lr_t = p.opt_lr, eps_t = 0.f;
    if (opt && adam) {
      const int t = (p.opt_step ? *p.opt_step : 0) + p.opt_step_offset;
      const float bc1 = 1.f - powf(p.opt_b1, (float)t), bc2 = 1.f - powf(p.opt_b2, (float)t);
      const float sq = sqrtf(bc2);
      lr_t = p.opt_lr * sq / bc1;
      eps_t = p.opt_eps * sq;
    }
#pragma unroll
    for (int ha = 0; ha < 2; ++ha) {
      // fp32 image of 64 rows x 64 columns: row R = 16i + 4g + r, 16-byte chunk (c >> 2) ^ ((R >> 2) & 15)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = hb * 32 + 16 * j + fr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int R = 16 * i + 4 * g + r;
              *reinterpret_cast<float*>(&wl[R * 256 + ((((c >> 2) ^ (R >> 2)) & 15) << 4) + (c & 3) * 4]) = acc[ha][i][hb][j][r];
            }
          }
      if (!opt) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int R = it * 4 + (lane >> 4), q = lane & 15;
          const float4 v = *reinterpret_cast<const float4*>(&wl[R * 256 + (((q ^ (R >> 2)) & 15) << 4)]);
          if (crow + ha * 64 + R < p.M && ccol + q * 4 < p.N)      // N is a multiple of 8: a chunk is inside or outside
            *reinterpret_cast<float4*>(&Cf[(size_t)(crow + ha * 64 + R) * p.ldc + ccol + q * 4]) = v;
        }
      } else {
        // two rows at a time (6 parameter loads in flight per lane before the first use); the fence keeps
        // the compiler from hoisting every group's loads to the top, which would not fit the register file.
        // A lane owns the ADJACENT rows 8ib + 2(lane >> 4) + {0, 1}, so that a pair of updated weights of one column
        // is one 32-bit piece of the transposed copy.
        unsigned short* const Wt = static_cast<unsigned short*>(p.opt_wt);
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) {
          __builtin_amdgcn_sched_barrier(0);
          f32x4 gq[2], pq[2], mq[2], vq[2];
          size_t off[2];
          bool in[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int R = ib * 8 + 2 * (lane >> 4) + u, q = lane & 15;
            gq[u] = *reinterpret_cast<const f32x4*>(&wl[R * 256 + (((q ^ (R >> 2)) & 15) << 4)]);
            in[u] = crow + ha * 64 + R < p.M && ccol + q * 4 < p.N;
            off[u] = in[u] ? (size_t)(crow + ha * 64 + R) * p.ldc + ccol + q * 4 : 0;
            pq[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.opt_p + off[u]));
            if (adam) {
              mq[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.opt_m + off[u]));
              vq[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.opt_v + off[u]));
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (adam) {
                float a = pq[u][e], b = mq[u][e], c = vq[u][e];
                dtf_adam_one(a, gq[u][e] * 1.0f, b, c, lr_t, p.opt_b1, p.opt_b2, eps_t);
                pq[u][e] = a; mq[u][e] = b; vq[u][e] = c;
              } else {
                pq[u][e] -= lr_t * gq[u][e];
              }
            }
            const unsigned w01 = (unsigned)pp_f2bf(pq[u][0]) | ((unsigned)pp_f2bf(pq[u][1]) << 16);
            const unsigned w23 = (unsigned)pp_f2bf(pq[u][2]) | ((unsigned)pp_f2bf(pq[u][3]) << 16);
            if (in[u]) {
              __builtin_nontemporal_store(pq[u], reinterpret_cast<f32x4*>(p.opt_p + off[u]));
              if (adam) {
                __builtin_nontemporal_store(mq[u], reinterpret_cast<f32x4*>(p.opt_m + off[u]));
                __builtin_nontemporal_store(vq[u], reinterpret_cast<f32x4*>(p.opt_v + off[u]));
              }
              if (p.opt_w) *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p.opt_w) + off[u]) = make_uint2(w01, w23);
            }
          }
          if (opt && Wt) {
            // transposed bf16 image, built in the part of the fp32 image this wave has already consumed (rows
            // 8ib..8ib+7 = bytes [2048 ib, 2048 ib + 2048) were read above; row group ib of the transposed image is
            // the 1 KB at 1024 ib): 16-byte chunk (ib, c) = rows 8ib..8ib+7 of column c, at slot
            // (c & 3)*16 + ((c >> 2) + 2 ib + (c & 1)) % 16, which keeps these stores and the reads below free of
            // bank conflicts.  LDS operations of one wave complete in order, and the region is private to the wave.
            const int lr = lane >> 4, q = lane & 15;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = 4 * q + e;
              const unsigned pair = (unsigned)pp_f2bf(pq[0][e]) | ((unsigned)pp_f2bf(pq[1][e]) << 16);
              *reinterpret_cast<unsigned*>(&wl[ib * 1024 + (((c & 3) << 4) + (((c >> 2) + 2 * ib + (c & 1)) & 15)) * 16 + lr * 4]) = pair;
            }
          }
        }
        if (opt && Wt) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int c = it * 8 + (lane >> 3), rg = lane & 7;
            const u32x4 v = *reinterpret_cast<const u32x4*>(&wl[rg * 1024 + (((c & 3) << 4) + (((c >> 2) + 2 * rg + (c & 1)) & 15)) * 16]);
            if (ccol + c < p.N && crow + ha * 64 + rg * 8 < p.M)      // M is a multiple of 8: a chunk is inside or outside
              *reinterpret_cast<u32x4*>(&Wt[(size_t)(ccol + c) * p.ldwt + crow + ha * 64 + rg * 8]) = v;
          }
        }
      }
    }
  } else {
    T* __restrict__ C = static_cast<T*>(p.C);
    T* __restrict__ Ct = static_cast<T*>(p.Ct);
    uint2 pk[2][4][2][2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = ccol + hb * 32 + 16 * j + fr;
        const bool col_ok = col < p.N;
        const float bias = (EPI == EPI_FWD && p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row0 = crow + ha * 64 + 16 * i + 4 * g;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ha][i][hb][j][r] + bias;
            if (EPI == EPI_FWD && p.relu) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (EPI == EPI_BWD) {
              // the lane's four rows are contiguous in the transposed mask (M is a multiple of 8: all four or none)
              const bool in = col_ok && row0 < p.M;
              const uint2 q = *reinterpret_cast<const uint2*>(static_cast<const T*>(p.maskT) + (in ? (size_t)col * p.ldmaskt + row0 : 0));
              const float mk[4] = {pp_bf2f((unsigned short)(q.x & 0xffff)), pp_bf2f((unsigned short)(q.x >> 16)),
                                   pp_bf2f((unsigned short)(q.y & 0xffff)), pp_bf2f((unsigned short)(q.y >> 16))};
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = mk[r] > 0.f ? v[r] : 0.f;
            }
            if (p.C32 && col_ok && row0 < p.M) {
#pragma unroll
              for (int r = 0; r < 4; ++r) p.C32[(size_t)(row0 + r) * p.ldc32 + col] = v[r];
            }
            pk[ha][i][hb][j].x = (unsigned)pp_f2bf(v[0]) | ((unsigned)pp_f2bf(v[1]) << 16);
            pk[ha][i][hb][j].y = (unsigned)pp_f2bf(v[2]) | ((unsigned)pp_f2bf(v[3]) << 16);
          }
      }
    if (Ct) {
      // transposed image [64 columns][128 rows] bf16: column c, 16-byte chunk (rows 8q..8q+7) at q ^ (c & 15)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ha = 0; ha < 2; ++ha)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int c = hb * 32 + 16 * j + fr, q = ha * 8 + 2 * i + (g >> 1);
              *reinterpret_cast<uint2*>(&wl[c * 256 + ((q ^ (c & 15)) << 4) + 8 * (g & 1)]) = pk[ha][i][hb][j];
            }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int c = it * 4 + (lane >> 4), q = lane & 15;
        const u32x4 v = *reinterpret_cast<const u32x4*>(&wl[c * 256 + ((q ^ (c & 15)) << 4)]);
        if (ccol + c < p.N && crow + q * 8 < p.M PP_DIAG_KEEP(2))
          *reinterpret_cast<u32x4*>(&Ct[(size_t)(ccol + c) * p.ldct + crow + q * 8]) = v;
      }
    }
    // row-major image [128 rows][64 columns] bf16: row R, 16-byte chunk (columns 8q..8q+7) at q ^ ((R >> 2) & 7)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = hb * 32 + 16 * j + fr;
            const uint2 v = pk[ha][i][hb][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int R = ha * 64 + 16 * i + 4 * g + r;
              const unsigned w32 = r < 2 ? v.x : v.y;
              *reinterpret_cast<unsigned short*>(&wl[R * 128 + ((((c >> 3) ^ (R >> 2)) & 7) << 4) + (c & 7) * 2]) =
                  (unsigned short)((r & 1) ? (w32 >> 16) : (w32 & 0xffff));
            }
          }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int R = it * 8 + (lane >> 3), q = lane & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(&wl[R * 128 + (((q ^ (R >> 2)) & 7) << 4)]);
      if (crow + R < p.M && ccol + q * 8 < p.N PP_DIAG_KEEP(1))
        *reinterpret_cast<u32x4*>(&C[(size_t)(crow + R) * p.ldc + ccol + q * 8]) = v;
    }
  }
  if (want_bias) {
    // lanes fr of the 4 lane groups hold disjoint k ranges of B row wc*64 + hb*32 + 16j + fr
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float s = bias_grad[hb][j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        float* const bo = (EPI == EPI_WGRAD && p.ksplit > 1) ? p.split_bias_ws + (size_t)split * p.N : p.bias_out;
        if (g == 0 && n0 + wc * 64 + hb * 32 + 16 * j + fr < p.N) bo[n0 + wc * 64 + hb * 32 + 16 * j + fr] = s;
      }
  }
}

// Shapes this kernel takes: bf16, M, N and K multiples of 8 (16-byte chunks of operands and outputs), K at least
// 256, the transposed mask for the backward epilogue, outputs that take 16-byte stores.  Whether it PAYS (enough
// 256x256 blocks to fill the chip, little waste at ragged edges) is the caller's decision.
// dst[i] = sum over the splits, in split order, of the partial weight gradients (ld = ldc: M * ldc floats per split,
// a multiple of 4) and partial bias gradients
__global__ __launch_bounds__(256) void splitk_sum_kernel(float* __restrict__ dst, const float* __restrict__ ws, long n4, long stride,
                                                         float* __restrict__ bias_dst, const float* __restrict__ bias_ws, int N, int S) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    f32x4 acc = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ws) + i);
    for (int s = 1; s < S; ++s) {
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ws + (size_t)s * stride) + i);
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    reinterpret_cast<f32x4*>(dst)[i] = acc;
  } else if (bias_dst && i - n4 < N) {
    const int c = (int)(i - n4);
    float acc = bias_ws[c];
    for (int s = 1; s < S; ++s) acc += bias_ws[(size_t)s * N + c];
    bias_dst[c] = acc;
  }
}

bool dtf_gemm_pingpong_takes(const DtfGemmArgs& a) {
  if (a.dtype != 1 || (a.M & 7) || (a.N & 7) || (a.K & 7) || a.K < 256) return false;
  if (a.ksplit > 1) {
    if (a.epilogue != EPI_WGRAD || a.opt_p || !a.split_ws || (a.bias_out && !a.split_bias_ws)) return false;
    if (a.ksplit > 16 || a.K / a.ksplit < 256 || (reinterpret_cast<uintptr_t>(a.split_ws) & 15)) return false;
    if (a.ldc != a.N) return false;           // the summing pass treats C and each partial as M * N contiguous floats
  }
  if (a.epilogue == EPI_BWD && (!a.maskT || (a.ldmaskt & 3))) return false;
  if ((reinterpret_cast<uintptr_t>(a.C) & 15) || (a.ldc & (a.epilogue == EPI_WGRAD ? 3 : 7))) return false;
  if (a.Ct && ((a.ldct & 7) || (reinterpret_cast<uintptr_t>(a.Ct) & 15))) return false;
  if (a.maskT && (reinterpret_cast<uintptr_t>(a.maskT) & 7)) return false;
  if (a.opt_p) {
    if (a.epilogue != EPI_WGRAD || (a.opt_m == nullptr) != (a.opt_v == nullptr)) return false;
    if ((reinterpret_cast<uintptr_t>(a.opt_p) & 15) || (a.opt_m && ((reinterpret_cast<uintptr_t>(a.opt_m) | reinterpret_cast<uintptr_t>(a.opt_v)) & 15))) return false;
    if (a.opt_w && (reinterpret_cast<uintptr_t>(a.opt_w) & 7)) return false;
    if (a.opt_wt && ((reinterpret_cast<uintptr_t>(a.opt_wt) & 15) || (a.ldwt & 7) || a.ldwt < a.M)) return false;
  }
  return true;
}

int dtf_gemm_pingpong_launch(const DtfGemmArgs& a_in, hipStream_t s) {
  if (!dtf_gemm_pingpong_takes(a_in)) return DTF_ERR_SHAPE;
#ifdef DTF_STAMPS
  static int skip = -1;
  if (skip < 0) {
    const char* e = getenv("DTF_GEMM_SKIP_STORES");
    skip = e ? atoi(e) & 3 : 0;
  }
  DtfGemmArgs a = a_in;
  a.pad_ = skip;
#else
  const DtfGemmArgs& a = a_in;
#endif
  const int split = a.ksplit > 1 ? a.ksplit : 1;
  dim3 grid(((a.M + 255) >> 8) * ((a.N + 255) >> 8) * split), block(512);
  switch (a.epilogue) {
    case EPI_FWD: hipLaunchKernelGGL(gemm_pp_kernel<EPI_FWD>, grid, block, 0, s, a); break;
    case EPI_BWD: hipLaunchKernelGGL(gemm_pp_kernel<EPI_BWD>, grid, block, 0, s, a); break;
    case EPI_WGRAD:
      if (a.opt_p) hipLaunchKernelGGL(gemm_pp_kernel<EPI_WGRAD_OPT>, grid, block, 0, s, a);
      else hipLaunchKernelGGL(gemm_pp_kernel<EPI_WGRAD>, grid, block, 0, s, a);
      break;
    default: return DTF_ERR_ARG;
  }
  if (split > 1) {
    const long n4 = (long)a.M * a.ldc / 4;
    const long threads = n4 + (a.bias_out ? a.N : 0);
    hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, static_cast<float*>(a.C),
                       a.split_ws, n4, (long)a.M * a.ldc, a.bias_out, a.split_bias_ws, a.N, split);
  }
  return dtf_check_launch();
}
