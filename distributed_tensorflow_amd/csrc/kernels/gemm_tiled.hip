// Throughput GEMM for the wide layers:  C[M,N] = A[M,K] . B[N,K]^T   (both operands k-contiguous)
//
// All three products of a dense layer are brought to this one form by keeping
// transposed copies where a product needs them (they are written by the
// producing kernel's epilogue, never by a separate transpose pass over HBM):
//
//   forward    Y  = X  . W        A = X [batch,in]      B = W^T [out,in]      (+bias, ReLU)
//   backward   dX = dY . W^T      A = dY [batch,out]    B = W   [in,out]      (* (act > 0))
//   gradient   dW = X^T . dY      A = X^T [in,batch]    B = dY^T [out,batch]  (+ db = rowsum(dY^T))
//
// This file: the register-staged 4-wave kernels, which take every shape (ragged edges through the range check of
// the buffer loads) and both dtypes; large aligned bf16 products go to the 8-wave kernel in gemm_pingpong.hip
// (dispatch in launch_tiled below).
// Tile: 128x128 per workgroup, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles of 16x16 (or 256x256 with 8x8
// tiles per wave, see the template parameters).
// K-step per LDS stage: 128 bytes per row (64 bf16 / 32 fp32), two stages (double buffer),
// register-staged: the global loads of stage t+1 are issued before the MFMAs of stage t and
// written to the other LDS buffer after them, one barrier per stage.
// LDS rows are 128 B; the 16-byte slot index is XOR-ed with (row & 7), which makes every
// ds_read_b128 fragment read conflict-free (the 16-lane groups of a b128 read cover 16
// distinct slots of the 256-byte bank row).
// bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  fp32: v_mfma_f32_16x16x4_f32 (exact fp32).
// Workgroup ids are remapped so that the workgroups sharing an XCD (ids equal mod 8) cover a
// contiguous band of tiles and share operand panels in that XCD's L2.
#include <cstdlib>

#include "dtf_kernels.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
static constexpr unsigned kOOB = 0xFFFFFFF0u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned short f2bf(float f) {  // round to nearest even; NaN stays NaN
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

enum { EPI_FWD = 0, EPI_BWD = 1, EPI_WGRAD = 2 };

template <typename T> struct Elem;
template <> struct Elem<unsigned short> { static constexpr int PER16 = 8; };   // bf16 elements per 16 bytes
template <> struct Elem<unsigned int> { static constexpr int PER16 = 4; };      // fp32 carried as its bit pattern

template <typename T>
__device__ __forceinline__ void mma_slot(const u32x4& a, const u32x4& b, f32x4& c);
template <>
__device__ __forceinline__ void mma_slot<unsigned short>(const u32x4& a, const u32x4& b, f32x4& c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_slot<unsigned int>(const u32x4& a, const u32x4& b, f32x4& c) {
  // lane group g holds k = 4g..4g+3 of this 16-deep slot; MFMA j consumes element j of every group
#pragma unroll
  for (int j = 0; j < 4; ++j)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
}

// TI x TJ: MFMA tiles per wave (wave tile 16*TI x 16*TJ); the 2x2 waves of a workgroup cover 32*TI x 32*TJ.
//   4 x 4: 128x128 block, 64 KB of LDS, two workgroups per CU; per 16 MFMAs a wave reads 8 fragments, which
//          keeps the LDS port saturated (measured: the loop is LDS-read bound).
//   8 x 8: 256x256 block, 128 KB of LDS, one workgroup per CU; 16 fragment reads per 64 MFMAs halve the LDS
//          traffic per flop.  Used for bf16 when the problem fills the chip with such blocks.
template <typename T, int EPI, int TI, int TJ>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(DtfGemmArgs p) {
  constexpr int PER16 = Elem<T>::PER16;
  constexpr int BKE = 8 * PER16;             // elements per 128-byte LDS row
  constexpr int BM = 32 * TI, BN = 32 * TJ;
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[2][BM * 128];   // [stage][row*128 + byte]
  __shared__ __attribute__((aligned(16))) unsigned char lds_b[2][BN * 128];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int wg = dtf_xcd_remap(blockIdx.x, tiles_m * tiles_n);
  // consecutive (remapped) ids walk down a column band of 8 row-tiles: A panels and one B panel stay in L2
  const int band = 8;
  const int per_band = band * tiles_n;
  const int b_idx = wg / per_band, in_band = wg % per_band;
  const int rows_in_band = min(band, tiles_m - b_idx * band);
  const int tm = b_idx * band + in_band % rows_in_band;
  const int tn = in_band / rows_in_band;
  const int m0 = tm * BM, n0 = tn * BN;

  const rsrc_t ra = make_rsrc(p.A, p.a_bytes), rb = make_rsrc(p.B, p.b_bytes);

  // staging map: BM*8 (A) and BN*8 (B) 16-byte chunks per stage, TI resp. TJ per thread; chunk = (row, slot)
  unsigned a_goff[TI], b_goff[TJ];
  unsigned la_off[TI], lb_off[TJ];
#pragma unroll
  for (int c = 0; c < TI; ++c) {
    const int id = tid + 256 * c;
    const int row = id >> 3, slot = id & 7;
    a_goff[c] = (m0 + row < p.M) ? (unsigned)(m0 + row) * (unsigned)p.lda * (unsigned)sizeof(T) + (unsigned)slot * 16u : kOOB;
    la_off[c] = (unsigned)row * 128u + (unsigned)((slot ^ (row & 7)) << 4);
  }
#pragma unroll
  for (int c = 0; c < TJ; ++c) {
    const int id = tid + 256 * c;
    const int row = id >> 3, slot = id & 7;
    b_goff[c] = (n0 + row < p.N) ? (unsigned)(n0 + row) * (unsigned)p.ldb * (unsigned)sizeof(T) + (unsigned)slot * 16u : kOOB;
    lb_off[c] = (unsigned)row * 128u + (unsigned)((slot ^ (row & 7)) << 4);
  }
  const int nstage = (p.K + BKE - 1) / BKE;
  const int my_slot = tid & 7;               // (tid + 256 c) & 7 for every c
  u32x4 ra_reg[TI], rb_reg[TJ];
  auto stage_load = [&](int t) {
    const int k0 = t * BKE;
    const bool kin = k0 + my_slot * PER16 < p.K;           // K is a multiple of PER16 (checked by the launcher)
    const unsigned kb = (unsigned)k0 * (unsigned)sizeof(T);
#pragma unroll
    for (int c = 0; c < TI; ++c)
      ra_reg[c] = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)((kin && a_goff[c] != kOOB) ? a_goff[c] + kb : kOOB), 0, 0);
#pragma unroll
    for (int c = 0; c < TJ; ++c)
      rb_reg[c] = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)((kin && b_goff[c] != kOOB) ? b_goff[c] + kb : kOOB), 0, 0);
  };
  auto stage_write = [&](int buf) {
#pragma unroll
    for (int c = 0; c < TI; ++c) *reinterpret_cast<u32x4*>(&lds_a[buf][la_off[c]]) = ra_reg[c];
#pragma unroll
    for (int c = 0; c < TJ; ++c) *reinterpret_cast<u32x4*>(&lds_b[buf][lb_off[c]]) = rb_reg[c];
  };

  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bias_grad[TJ];
#pragma unroll
  for (int j = 0; j < TJ; ++j) bias_grad[j] = 0.f;

  // fragment read offsets: row r = wave base + 16 i + fr, slot s = 4 ks + g  ->  r*128 + ((s ^ (r & 7)) << 4)
  const unsigned a_rbase = (unsigned)(wm * 16 * TI + fr) * 128u, b_rbase = (unsigned)(wn * 16 * TJ + fr) * 128u;
  const int sw = fr & 7;     // (row & 7) == (fr & 7): the row bases are multiples of 16

  auto read_frags = [&](int buf, int ks, u32x4 (&af)[TI], u32x4 (&bf)[TJ]) {
    const unsigned so = (unsigned)(((4 * ks + g) ^ sw) << 4);
#pragma unroll
    for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const u32x4*>(&lds_b[buf][b_rbase + (unsigned)j * 2048u + so]);
#pragma unroll
    for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const u32x4*>(&lds_a[buf][a_rbase + (unsigned)i * 2048u + so]);
  };
  auto bias_accumulate = [&](const u32x4 (&bf)[TJ]) {
    // db[n] = sum_k dY^T[n][k]: each lane adds the 16 bytes it holds of its B rows
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      if (sizeof(T) == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_grad[j] += bf2f((unsigned short)(bf[j][e] & 0xffff)) + bf2f((unsigned short)(bf[j][e] >> 16));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) bias_grad[j] += __uint_as_float(bf[j][e]);
      }
    }
  };

  stage_load(0);
  stage_write(0);
  __syncthreads();
  if (TI >= 8) {
    // One wave per SIMD: nothing else hides an LDS read, so the fragments of the second k-step are requested
    // before the MFMAs of the first (two fragment sets in registers), and the next stage's operands are
    // written to the other buffer between the two k-steps.
    u32x4 af0[TI], bf0[TJ], af1[TI], bf1[TJ];
    for (int t = 0; t < nstage; ++t) {
      const int cur = t & 1;
      const bool more = t + 1 < nstage;
      if (more) stage_load(t + 1);
      read_frags(cur, 0, af0, bf0);
      read_frags(cur, 1, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      if (EPI == EPI_WGRAD) bias_accumulate(bf0);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) mma_slot<T>(af0[i], bf0[j], acc[i][j]);
      if (more) stage_write(cur ^ 1);          // last read a barrier ago
      if (EPI == EPI_WGRAD) bias_accumulate(bf1);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) mma_slot<T>(af1[i], bf1[j], acc[i][j]);
      __syncthreads();
    }
  } else {
    for (int t = 0; t < nstage; ++t) {
      const int cur = t & 1;
      if (t + 1 < nstage) stage_load(t + 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 af[TI], bf[TJ];
        read_frags(cur, ks, af, bf);
        if (EPI == EPI_WGRAD) bias_accumulate(bf);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) mma_slot<T>(af[i], bf[j], acc[i][j]);
      }
      if (t + 1 < nstage) stage_write(cur ^ 1);
      __syncthreads();
    }
  }

  // ---------------- epilogue: acc[i][j][r] = C[m0 + wm*16*TI + 16i + 4g + r][n0 + wn*16*TJ + 16j + fr]
  T* __restrict__ C = static_cast<T*>(p.C);
  T* __restrict__ Ct = static_cast<T*>(p.Ct);
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int col = n0 + wn * 16 * TJ + 16 * j + fr;
    const bool col_ok = col < p.N;
    const float bias = (EPI == EPI_FWD && p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const int row0 = m0 + wm * 16 * TI + 16 * i + 4 * g;
      float v[4];
      float mk[4] = {1.f, 1.f, 1.f, 1.f};
      if (EPI == EPI_BWD && col_ok) {
        if (p.maskT && row0 + 3 < p.M && (p.ldmaskt & 3) == 0) {
          // the lane's four rows are contiguous in the transposed mask: one load instead of four
          const T* mt = static_cast<const T*>(p.maskT) + (size_t)col * p.ldmaskt + row0;
          if (sizeof(T) == 2) {
            const uint2 q = *reinterpret_cast<const uint2*>(mt);
            mk[0] = bf2f((unsigned short)(q.x & 0xffff)); mk[1] = bf2f((unsigned short)(q.x >> 16));
            mk[2] = bf2f((unsigned short)(q.y & 0xffff)); mk[3] = bf2f((unsigned short)(q.y >> 16));
          } else {
            const float4 q = *reinterpret_cast<const float4*>(mt);
            mk[0] = q.x; mk[1] = q.y; mk[2] = q.z; mk[3] = q.w;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) {
              const T m = static_cast<const T*>(p.mask)[(size_t)(row0 + r) * p.ldmask + col];
              mk[r] = sizeof(T) == 2 ? bf2f((unsigned short)m) : __uint_as_float((unsigned)m);
            }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + bias;
        if (EPI == EPI_FWD && p.relu) v[r] = fmaxf(v[r], 0.f);
        if (EPI == EPI_BWD) v[r] = mk[r] > 0.f ? v[r] : 0.f;
      }
      if (!col_ok) continue;
      if (EPI == EPI_WGRAD) {
        float* __restrict__ Cf = static_cast<float*>(p.C);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (row0 + r < p.M) Cf[(size_t)(row0 + r) * p.ldc + col] = v[r];
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (row0 + r < p.M) {
          if (sizeof(T) == 2) C[(size_t)(row0 + r) * p.ldc + col] = (T)f2bf(v[r]);
          else C[(size_t)(row0 + r) * p.ldc + col] = (T)__float_as_uint(v[r]);
          if (p.C32) p.C32[(size_t)(row0 + r) * p.ldc32 + col] = v[r];
        }
      }
      if (Ct) {
        // transposed copy: the 4 rows a lane holds are contiguous in Ct's row `col`
        if (row0 + 3 < p.M && (p.ldct & 3) == 0) {
          if (sizeof(T) == 2) {
            uint2 pk;
            pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
            pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(&Ct[(size_t)col * p.ldct + row0]) = pk;
          } else {
            *reinterpret_cast<float4*>(&Ct[(size_t)col * p.ldct + row0]) = make_float4(v[0], v[1], v[2], v[3]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < p.M) {
              if (sizeof(T) == 2) Ct[(size_t)col * p.ldct + row0 + r] = (T)f2bf(v[r]);
              else Ct[(size_t)col * p.ldct + row0 + r] = (T)__float_as_uint(v[r]);
            }
        }
      }
    }
  }
  if (EPI == EPI_WGRAD && p.bias_out && tm == 0 && wm == 0) {
    // lanes fr of the 4 lane groups hold disjoint k ranges of B row (wn*16*TJ + 16i + fr)
#pragma unroll
    for (int i = 0; i < TJ; ++i) {
      float s = bias_grad[i];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const int col = n0 + wn * 16 * TJ + 16 * i + fr;
      if (g == 0 && col < p.N) p.bias_out[col] = s;
    }
  }
}

template <typename T, int TI, int TJ>
static int launch_tiled_as(const DtfGemmArgs& a, hipStream_t s) {
  const int tiles = dtf_cdiv(a.M, 32 * TI) * dtf_cdiv(a.N, 32 * TJ);
  dim3 grid(tiles), block(256);
  switch (a.epilogue) {
    case EPI_FWD: hipLaunchKernelGGL((gemm_tiled_kernel<T, EPI_FWD, TI, TJ>), grid, block, 0, s, a); break;
    case EPI_BWD: hipLaunchKernelGGL((gemm_tiled_kernel<T, EPI_BWD, TI, TJ>), grid, block, 0, s, a); break;
    case EPI_WGRAD: hipLaunchKernelGGL((gemm_tiled_kernel<T, EPI_WGRAD, TI, TJ>), grid, block, 0, s, a); break;
    default: return DTF_ERR_ARG;
  }
  return dtf_check_launch();
}

// gemm_pingpong.hip: 8-wave 256x256 kernel for large aligned bf16 products
bool dtf_gemm_pingpong_takes(const DtfGemmArgs& a);
int dtf_gemm_pingpong_launch(const DtfGemmArgs& a, hipStream_t s);
static int g_pingpong_mode = -1;      // -1 = unread, 0 = never, 1 = whenever the shape allows, 2 = by problem size (DTF_GEMM_PINGPONG)

static int g_big_tile_mode = -1;      // -1 = by problem size, 0 = never, 1 = always (DTF_GEMM_BIG_TILE)

template <typename T>
static int launch_tiled(const DtfGemmArgs& a, hipStream_t s) {
  if (g_big_tile_mode < 0) {
    const char* e = getenv("DTF_GEMM_BIG_TILE");
    g_big_tile_mode = e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2)) : 2;
  }
  // 256x256 blocks pay off once they fill the chip (one block per CU, 256 CUs) and waste little at the edges
  const long big_tiles = (long)dtf_cdiv(a.M, 256) * dtf_cdiv(a.N, 256);
  const bool fits = big_tiles >= 192 && (a.M % 256 == 0 || a.M >= 2048) && (a.N % 256 == 0 || a.N >= 2048);
  // measured at 8192x4096x4096 bf16: forward 714 -> 787, weight gradient 778 -> 901 TFLOP/s; the backward epilogue
  // (one mask element per output element) is not overlapped by a second resident workgroup any more: 663 -> 587,
  // so it keeps the 128x128 blocks
  const bool suited = a.epilogue != EPI_BWD || a.maskT != nullptr;
  if (g_pingpong_mode < 0) {
    const char* e = getenv("DTF_GEMM_PINGPONG");
    g_pingpong_mode = e ? (e[0] == '1' ? 1 : (e[0] == '0' ? 0 : 2)) : 2;
  }
  if (a.opt_p || a.ksplit > 1)      // folded optimizer, split-K: 8-wave kernel only, and the caller asked for exactly that
    return (sizeof(T) == 2 && dtf_gemm_pingpong_takes(a)) ? dtf_gemm_pingpong_launch(a, s) : (int)DTF_ERR_SHAPE;
  if (sizeof(T) == 2 && g_pingpong_mode && dtf_gemm_pingpong_takes(a) && (g_pingpong_mode == 1 || fits))
    return dtf_gemm_pingpong_launch(a, s);
  if (sizeof(T) == 2 && (g_big_tile_mode == 1 || (g_big_tile_mode == 2 && fits && suited))) return launch_tiled_as<T, 8, 8>(a, s);
  return launch_tiled_as<T, 4, 4>(a, s);
}

// Kernel selection for tests and tools: 0 = never, 1 = whenever the shape allows, 2 = by problem size, -1 = leave as is
// (the defaults come from DTF_GEMM_BIG_TILE / DTF_GEMM_PINGPONG, read on first use).
DTF_EXPORT void dtf_gemm_set_modes(int big_tile, int pingpong) {
  if (big_tile >= 0) g_big_tile_mode = big_tile;
  if (pingpong >= 0) g_pingpong_mode = pingpong;
}

DTF_EXPORT int dtf_gemm_tiled(const DtfGemmArgs* a, hipStream_t s) {
  if (!a || !a->A || !a->B || !a->C) return DTF_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return DTF_ERR_SHAPE;
  const int per16 = a->dtype == 1 ? 8 : 4;
  if (a->dtype != 0 && a->dtype != 1) return DTF_ERR_ARG;
  // 16-byte staging loads: K and both leading dimensions in whole 16-byte chunks, aligned bases
  if (a->K % per16 || a->lda % per16 || a->ldb % per16) return DTF_ERR_ALIGN;
  if (!dtf_aligned16(a->A) || !dtf_aligned16(a->B)) return DTF_ERR_ALIGN;
  if (a->lda < a->K || a->ldb < a->K || a->ldc < a->N) return DTF_ERR_SHAPE;
  const uint64_t es = a->dtype == 1 ? 2 : 4;
  if (a->a_bytes == 0 || a->b_bytes == 0 || a->a_bytes >= kOOB || a->b_bytes >= kOOB) return DTF_ERR_SHAPE;
  if (((uint64_t)(a->M - 1) * a->lda + a->K) * es > a->a_bytes) return DTF_ERR_SHAPE;
  if (((uint64_t)(a->N - 1) * a->ldb + a->K) * es > a->b_bytes) return DTF_ERR_SHAPE;
  if (a->epilogue == EPI_BWD && (!a->mask || a->ldmask < a->N)) return DTF_ERR_ARG;
  if (a->maskT && (a->ldmaskt < a->M || !dtf_aligned16(a->maskT))) return DTF_ERR_SHAPE;
  if (a->Ct && a->ldct < a->M) return DTF_ERR_SHAPE;
  if (a->C32 && a->ldc32 < a->N) return DTF_ERR_SHAPE;
  if (a->epilogue == EPI_WGRAD && (a->Ct || a->C32)) return DTF_ERR_ARG;
  if (a->opt_p && a->epilogue != EPI_WGRAD) return DTF_ERR_ARG;
  return a->dtype == 1 ? launch_tiled<unsigned short>(*a, s) : launch_tiled<unsigned int>(*a, s);
}

// --------------------------------------------------------------------------------------
// helpers of the wide pipeline (pure streaming kernels)
// --------------------------------------------------------------------------------------
// dst[i] = src[rows[i]] converted to T, plus the transposed copy dstT[c][i]  (batch gather for layer 1,
// compute copies of the weights).  64x64 tiles: 16-byte global loads, the tile held in LDS in the output
// type, straight and transposed copies written as 8-byte (bf16) / 16-byte (fp32) pieces of 4 elements.
template <typename T>
__global__ __launch_bounds__(256) void gather_cast_transpose_kernel(const float* __restrict__ src, const int* __restrict__ rows,
                                                                    T* __restrict__ dst, T* __restrict__ dstT, int M, int N,
                                                                    int lds_, int ldd, int ldt) {
  __shared__ T tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, 4 elements each per pass
  const bool vec_in = ((lds_ & 3) == 0) && (((uintptr_t)src & 15) == 0);
  const bool vec_out = dst && ((ldd & 3) == 0) && (((uintptr_t)dst & 15) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = ty + 16 * i, r = r0 + rr, c = c0 + 4 * tx;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < M) {
      const size_t base = (size_t)(rows ? rows[r] : r) * lds_ + c;
      if (vec_in && c + 3 < N) {
        const float4 q = *reinterpret_cast<const float4*>(src + base);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < N) v[j] = src[base + j];
      }
    }
    T o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = sizeof(T) == 2 ? (T)f2bf(v[j]) : (T)__float_as_uint(v[j]);
      tile[rr][4 * tx + j] = o[j];
    }
    if (dst && r < M) {
      T* d = dst + (size_t)r * ldd + c;
      if (vec_out && c + 3 < N) {
        if (sizeof(T) == 2) *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
        else *reinterpret_cast<uint4*>(d) = make_uint4((unsigned)o[0], (unsigned)o[1], (unsigned)o[2], (unsigned)o[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < N) d[j] = o[j];
      }
    }
  }
  __syncthreads();
  if (!dstT) return;
  const bool vec_t = ((ldt & 3) == 0) && (((uintptr_t)dstT & 15) == 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cc = ty + 16 * i, c = c0 + cc, r = r0 + 4 * tx;   // output row c, 4 consecutive source rows
    if (c >= N) continue;
    T o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = tile[4 * tx + j][cc];
    T* d = dstT + (size_t)c * ldt + r;
    if (vec_t && r + 3 < M) {
      if (sizeof(T) == 2) *reinterpret_cast<uint2*>(d) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
      else *reinterpret_cast<uint4*>(d) = make_uint4((unsigned)o[0], (unsigned)o[1], (unsigned)o[2], (unsigned)o[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (r + j < M) d[j] = o[j];
    }
  }
}

// dstT[c][r] = src[r][c] for 2-byte elements (the transposed compute copy of weights whose straight copy the
// weight-gradient epilogue has just written).  64x64 tiles through LDS, 16-byte global accesses on both sides.
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dstT,
                                                          int M, int N, int lds_, int ldt) {
  __shared__ unsigned short tile[64][72];           // 144-byte rows: 16-byte aligned, column reads spread over banks
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = t + 256 * i, rr = id >> 3, ch = id & 7;          // row, 16-byte chunk of 8 columns
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + rr < M && c0 + ch * 8 < N) v = *reinterpret_cast<const u32x4*>(src + (size_t)(r0 + rr) * lds_ + c0 + ch * 8);
    *reinterpret_cast<u32x4*>(&tile[rr][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = t + 256 * i, cc = id >> 3, ch = id & 7;          // output row (a source column), chunk of 8 source rows
    if (c0 + cc >= N || r0 + ch * 8 >= M) continue;
    unsigned short e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = tile[ch * 8 + k][cc];
    u32x4 v = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
               (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
    *reinterpret_cast<u32x4*>(dstT + (size_t)(c0 + cc) * ldt + r0 + ch * 8) = v;
  }
}

DTF_EXPORT int dtf_transpose16(const void* src, void* dstT, int M, int N, int lds_, int ldt, hipStream_t s) {
  if (!src || !dstT) return DTF_ERR_ARG;
  if (M <= 0 || N <= 0 || (M & 7) || (N & 7) || lds_ < N || ldt < M || (lds_ & 7) || (ldt & 7)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(src) || !dtf_aligned16(dstT)) return DTF_ERR_ALIGN;
  if (dtf_cdiv(M, 64) > 65535) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(transpose16_kernel, dim3(dtf_cdiv(N, 64), dtf_cdiv(M, 64)), dim3(256), 0, s,
                     static_cast<const unsigned short*>(src), static_cast<unsigned short*>(dstT), M, N, lds_, ldt);
  return dtf_check_launch();
}

DTF_EXPORT int dtf_gather_cast_transpose(const float* src, const int* rows, void* dst, void* dstT, int M, int N, int lds_,
                                         int ldd, int ldt, int dtype, hipStream_t s) {
  if (!src || (!dst && !dstT)) return DTF_ERR_ARG;
  if (M <= 0 || N <= 0 || lds_ < N || (dst && ldd < N) || (dstT && ldt < M)) return DTF_ERR_SHAPE;
  if (dtf_cdiv(M, 64) > 65535) return DTF_ERR_SHAPE;
  dim3 grid(dtf_cdiv(N, 64), dtf_cdiv(M, 64)), block(256);
  if (dtype == 1)
    hipLaunchKernelGGL(gather_cast_transpose_kernel<unsigned short>, grid, block, 0, s, src, rows, (unsigned short*)dst,
                       (unsigned short*)dstT, M, N, lds_, ldd, ldt);
  else if (dtype == 0)
    hipLaunchKernelGGL(gather_cast_transpose_kernel<unsigned int>, grid, block, 0, s, src, rows, (unsigned int*)dst,
                       (unsigned int*)dstT, M, N, lds_, ldd, ldt);
  else
    return DTF_ERR_ARG;
  return dtf_check_launch();
}
