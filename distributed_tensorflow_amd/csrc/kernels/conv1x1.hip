// Weight gradient of a stride-1 1x1 convolution on channels-last bf16 activations:
//
//     dW[co][ci] = sum over m of dy[m][co] * x[m][ci],        m = (image, row, column): up to 803 k rows in ResNet-50,
//
// a product with a tiny output (64 x 64 ... 2048 x 512) and a very long reduction whose two operands both have the
// REDUCTION index as their slow dimension.  An MFMA operand wants 8 consecutive reduction indices of one channel in
// a lane, i.e. a column of the [m][channel] tile: the tiles go into LDS row-major as they come from memory
// (buffer_load ... lds, 16 bytes per lane, no VGPR destination) and are read back with ds_read_b64_tr_b16, gfx950's
// transposing LDS read (per 16 lanes a block of 4 rows x 16 columns, delivered column-major).  The kernel is a
// streaming kernel - 64 bytes of operands per 128 x 128 x 1 slice of MACs - so the grid splits the reduction over as
// many workgroups as it takes to fill the chip a few times; partial results go to a workspace and a second launch
// adds them in split order (the result does not depend on the order in which workgroups finish) and rounds to bf16.
//
// Workgroup: 4 waves, output tile 128 (co) x 128 (ci), wave (wr, wc) owns 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
// (v_mfma_f32_32x32x16_bf16).  A step covers 16 rows of m: two [16][128] bf16 tiles (dy, x) of 4 KB each; eight such
// stages rotate through 64 KB of LDS, six are in flight while one is consumed (a step is one MFMA k-step per tile:
// what bounds the kernel is how many bytes a CU keeps in flight), one barrier per step.
// LDS image of a tile: 256-byte rows, 16-byte chunk ch of row r at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))):
// conflict-free for the transposed reads; the swizzle is applied on the SOURCE address of the lane-linear DMA.
#include "dtf_kernels.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
static constexpr unsigned kOOB = 0xFFFFFFF0u;

namespace {
constexpr int kStages = 8;
constexpr int kAhead = 6;                              // stages requested ahead of the one being consumed
constexpr int kRows = 16;                              // rows of m per stage
constexpr int kTileBytes = kRows * 256;                // one operand tile: 16 rows x 128 channels bf16

__device__ __forceinline__ unsigned swz(int row) { return (unsigned)(((row & 3) << 2) | ((row >> 2) & 3)); }
__device__ __forceinline__ unsigned tile_off(int row, int ch) { return 256u * (unsigned)row + 16u * ((unsigned)ch ^ swz(row)); }

// 8 consecutive reduction indices (rows r0 .. r0 + 7 of the tile) of the 16 channels 8 c0 .. 8 c0 + 15: lane t of a
// 16-lane group gets channel 8 c0 + t.  Two transposed reads of 4 rows each.
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* tile, int r0, int c0, int tl) {
  const int q = tl >> 2, p = tl & 3;
  const unsigned a0 = tile_off(r0 + q, c0 + (p >> 1)) + 8u * (unsigned)(p & 1);
  const unsigned a1 = tile_off(r0 + 4 + q, c0 + (p >> 1)) + 8u * (unsigned)(p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + a1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                            float* __restrict__ ws, long long M, int Cin, int Cout, int ksplit,
                                                            int rows_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kStages * 2 * kTileBytes];   // [stage][dy | x][16 rows x 256 B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_ci = (Cin + 127) >> 7, tiles_co = (Cout + 127) >> 7;
  const int tiles = tiles_ci * tiles_co;
  const int split = (int)blockIdx.x / tiles;
  const int t = (int)blockIdx.x - split * tiles;
  const int co0 = (t / tiles_ci) << 7, ci0 = (t % tiles_ci) << 7;
  const long long m_begin = (long long)split * rows_per_split;
  const long long m_end = m_begin + rows_per_split < M ? m_begin + rows_per_split : M;
  const int nk = m_end > m_begin ? (int)((m_end - m_begin + kRows - 1) / kRows) : 0;

  // one resource per operand covering this split's rows: offsets stay below 4 GB whatever M is
  const unsigned short* xs = x + (size_t)m_begin * Cin;
  const unsigned short* ds = dy + (size_t)m_begin * Cout;
  const long long rows_here = m_end > m_begin ? m_end - m_begin : 0;
  const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xs), 0, (int)(rows_here * Cin * 2), 0x00020000);
  const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(ds), 0, (int)(rows_here * Cout * 2), 0x00020000);

  // ---- staging: one wave-instruction fills 4 rows (1 KB) of a tile; wave w fills row group w of both tiles.
  //      Lane L writes physical chunk L & 15 of row 4 w + (L >> 4): it fetches logical chunk (L & 15) ^ swz(row).
  const int srow = 4 * wave + (lane >> 4);
  const int sch = (int)((unsigned)(lane & 15) ^ swz(srow));
  const unsigned x_off = (ci0 + 8 * sch < Cin) ? ((unsigned)srow * (unsigned)Cin + (unsigned)(ci0 + 8 * sch)) * 2u : kOOB;
  const unsigned d_off = (co0 + 8 * sch < Cout) ? ((unsigned)srow * (unsigned)Cout + (unsigned)(co0 + 8 * sch)) * 2u : kOOB;
  const unsigned x_step = (unsigned)kRows * (unsigned)Cin * 2u, d_step = (unsigned)kRows * (unsigned)Cout * 2u;
  auto stage = [&](int kt) {
    lds_byte_t* dst = (lds_byte_t*)(lds) + (kt & (kStages - 1)) * 2 * kTileBytes + wave * 1024;
    // rows past the split's end: out of range -> zeros (the range check covers the whole split, so the row test is explicit)
    const bool live = kt < nk && (long long)kt * kRows + srow < rows_here;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, dst, 16, (int)((live && d_off != kOOB) ? d_off + (unsigned)kt * d_step : kOOB), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst + kTileBytes, 16, (int)((live && x_off != kOOB) ? x_off + (unsigned)kt * x_step : kOOB), 0, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: lane -> (column of the 32-wide MFMA tile, half of the 16-deep k-step)
  const int col32 = lane & 31, kg = lane >> 5;
  const int gq = col32 >> 4, tl = lane & 15;

#pragma unroll
  for (int k = 0; k < kAhead; ++k) stage(k);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt + 6 goes where stage kt - 2 was: every wave finished reading that one before it arrived at the barrier
    // of step kt - 1, which this wave has passed
    stage(kt + kAhead);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // the two loads of stage kt have landed (12 newer ones may be in flight)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* td = lds + (kt & (kStages - 1)) * 2 * kTileBytes;
    const unsigned char* tx = td + kTileBytes;
    const int r0 = 8 * kg;
    bf16x8_t a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = tr_frag(td, r0, wr * 8 + i * 4 + 2 * gq, tl);
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = tr_frag(tx, r0, wc * 8 + j * 4 + 2 * gq, tl);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing may land in LDS after the workgroup has ended

  // ---- epilogue: acc[i][j][r] = dW[co0 + wr*64 + 32i + 8 (r >> 2) + 4 kg + (r & 3)][ci0 + wc*64 + 32j + col32]
  float* out = ws + (size_t)split * Cout * Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = ci0 + wc * 64 + 32 * j + col32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wr * 64 + 32 * i + 8 * (r >> 2) + 4 * kg + (r & 3);
        if (co < Cout && ci < Cin) out[(size_t)co * Cin + ci] = acc[i][j][r];
      }
    }
}

// dW (bf16, [Cout][Cin]) = sum over the splits of the partial results, always in the same order: 16 lanes per four
// elements, lane l adds splits l, l + 16, ... in order, then the 16 sums are added in lane order.
__global__ __launch_bounds__(256) void conv1x1_wgrad_sum_kernel(const float* __restrict__ ws, unsigned short* __restrict__ dw, long long n, int ksplit) {
  const long long item = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const long long i = item * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    for (int s = l; s < ksplit; s += 16) {
      const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ws + (size_t)s * n + i));
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
  }
  // lane order: ((0+1)+(2+3)) + ... as a fixed tree over the 16 lanes
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 16);
  }
  if (i >= n || l != 0) return;
  const unsigned lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[1]) << 16);
  const unsigned hi = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[3]) << 16);
  *reinterpret_cast<uint2*>(dw + i) = make_uint2(lo, hi);
}

// how the reduction is split: 1024 workgroups where the rows allow it (two per CU are resident: two even rounds over
// 256 CUs), at least 24 steps of 16 rows each
int pick_split(long long M, int tiles) {
  long long want = 1024 / tiles;
  const long long most = (M + 24 * kRows - 1) / (24 * kRows);
  if (want > most) want = most;
  if (want < 1) want = 1;
  if (want > 1024) want = 1024;
  return (int)want;
}
}  // namespace

// Floats of workspace a call with these sizes needs.
DTF_EXPORT long long dtf_conv1x1_wgrad_ws_floats(long long M, int Cin, int Cout) {
  if (M <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int tiles = dtf_cdiv(Cin, 128) * dtf_cdiv(Cout, 128);
  return (long long)pick_split(M, tiles) * Cin * Cout;
}

// dw [Cout][Cin] bf16 = dy[M][Cout]^T x[M][Cin] (bf16, row-major, 16-byte aligned; Cin, Cout multiples of 8).
DTF_EXPORT int dtf_conv1x1_wgrad(const void* x, const void* dy, float* ws, void* dw, long long M, int Cin, int Cout, hipStream_t s) {
  if (!x || !dy || !ws || !dw) return DTF_ERR_ARG;
  if (M <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 7)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(x) || !dtf_aligned16(dy) || !dtf_aligned16(ws) || ((uintptr_t)dw & 7)) return DTF_ERR_ALIGN;
  if (((long long)Cin * Cout) & 3) return DTF_ERR_SHAPE;
  const int tiles = dtf_cdiv(Cin, 128) * dtf_cdiv(Cout, 128);
  const int ksplit = pick_split(M, tiles);
  long long rows = (M + ksplit - 1) / ksplit;
  rows = (rows + kRows - 1) / kRows * kRows;
  // a split's operands are addressed with 32-bit byte offsets
  if (rows * (long long)(Cin > Cout ? Cin : Cout) * 2 >= 0x7fffffffll) return DTF_ERR_SHAPE;
  hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3(tiles * ksplit), dim3(256), 0, s, static_cast<const unsigned short*>(x),
                     static_cast<const unsigned short*>(dy), ws, M, Cin, Cout, ksplit, (int)rows);
  const long long n = (long long)Cin * Cout;
  hipLaunchKernelGGL(conv1x1_wgrad_sum_kernel, dim3((unsigned)((n / 4 * 16 + 255) / 256)), dim3(256), 0, s, ws,
                     static_cast<unsigned short*>(dw), n, ksplit);
  return dtf_check_launch();
}
