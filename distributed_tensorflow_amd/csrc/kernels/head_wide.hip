// Output layer of a WIDE network in the bf16 pipeline (thousands of hidden units, at most 16 classes): the two
// kernels around the 10-class layer, reading the last hidden activation as the bf16 tensors the forward GEMM has
// already written (row-major H and transposed H^T) instead of an fp32 copy.
//
//   head_wide_kernel       logits, (double) softmax cross-entropy, accuracy count, dz2, and dz of the last hidden
//                          layer as bf16 row-major + transposed (the operands of the backward GEMMs)
//                          (distributed.py:81-87 of the reference: softmax, loss, argmax accuracy)
//   out_wgrad_wide_kernel  dW2 = H^T dz2, db2 = colsum(dz2), plus the step tail in one extra workgroup
//
// Both are latency kernels, not throughput kernels: 16 batch rows (or 16 hidden units) per workgroup give 256
// workgroups for a 4096-row batch, one per CU, so the time is the number of DEPENDENT memory passes per wave.
// A workgroup therefore has 16 waves that split the long dimension (4096 hidden units / batch rows) into contiguous
// sixteenths: two passes of four 16-byte activation loads per lane in the products, two passes of eight column
// chunks in the dz phase.  The 4-wave kernels of mlp_small.hip, built for the 100-unit reference layer, need 8
// passes per phase on this shape and read the activation as fp32.
//
// Measured at batch 4096 x 4096 hidden units (tools/head_wide_probe.py): head 44 us with operands in the caches, 70 us
// behind a 256 MB write (the 4-wave fp32 kernel: 60 / 70); logits + loss part 17 us, the dz part 27 us, of which the
// accesses to the TRANSPOSED tensors (mask from H^T, dz^T out: 32-byte pieces of 128-byte lines, because a workgroup
// owns only 16 batch rows) are 9 us warm and 28 us cold.  Weight gradient 12 us warm / 20 us cold.
//
// Workgroup ids are remapped so that consecutive row blocks run on the same XCD: four of them share every 128-byte
// line of H^T (64 batch rows of one hidden unit) and of the transposed dz output.
#include "dtf_device.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float hw_bf2f(unsigned h16) { return __uint_as_float(h16 << 16); }
__device__ __forceinline__ unsigned short hw_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ u32x4_t hw_ld16(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); }
__device__ __forceinline__ u32x2_t hw_ld8(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0); }

template <int NW>
__global__ __launch_bounds__(NW * 64) void head_wide_kernel(DtfHeadWideArgs p) {
  __shared__ float red[NW][4][64];
  __shared__ float dzs[16][17];
  __shared__ float part_loss[4];
  __shared__ int part_cnt[4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int blk = dtf_xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = blk * 16;
  const int C = p.C, Hd = p.Hd, M = p.M;
  const float NEG_INF = -__builtin_huge_valf();
  const bool cls_ok = fr < C;
  const rsrc_t rh = make_rsrc(p.H, (unsigned)(((size_t)(M - 1) * p.ldh + Hd) * 2u));
  const rsrc_t rht = make_rsrc(p.Ht, (unsigned)(((size_t)(Hd - 1) * p.ldht + M) * 2u));
  const rsrc_t rw = make_rsrc(p.W2, (unsigned)Hd * (unsigned)C * 4u);

  // what phase B needs from memory is requested before phase A (the label row is a dependent load): it has arrived by
  // the time the logits are complete
  const int brow = r0 + 4 * g + wave;                   // the row this lane group handles in phase B (waves 0..3)
  const bool brow_ok = wave < 4 && brow < M;
  float labv = 0.f, biasv = 0.f;
  if (wave < 4) {
    const int lrow = p.rows ? p.rows[brow_ok ? brow : 0] : brow;
    labv = p.labels[(size_t)(brow_ok ? lrow : 0) * C + (cls_ok ? fr : 0)];
    biasv = p.b2[cls_ok ? fr : 0];
  }

  // ---- phase A: partial logits of rows r0..r0+15 over this wave's sixteenth of the hidden units.  MFMA k-slice j of
  //      a 32-unit chunk pairs A[row fr][k0 + 8g + j] with W2[k0 + 8g + j][class fr]: a lane's eight activations are
  //      one 16-byte load.
  {
    const int nch = (Hd + 31) >> 5;
    const int cpw = (nch + NW - 1) / NW;
    const int c_lo = wave * cpw, c_hi = min(nch, c_lo + cpw);
    const int arow = r0 + fr;
    const bool arow_ok = arow < M;
    const unsigned a_row_off = (unsigned)arow * (unsigned)p.ldh * 2u;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    constexpr int CH = 8;                                // 4096 hidden units: one pass per wave
    for (int cb = c_lo; cb < c_hi; cb += CH) {
      u32x4_t hv[CH];
      float bv[CH][8];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int k = ((cb + i) << 5) + 8 * g;
        const bool live = cb + i < c_hi && k < Hd;          // Hd is a multiple of 8: eight units are inside or outside
        hv[i] = hw_ld16(rh, (live && arow_ok) ? a_row_off + (unsigned)k * 2u : kOOB);
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[i][j] = bld(rw, (live && cls_ok) ? ((unsigned)(k + j) * (unsigned)C + (unsigned)fr) * 4u : kOOB);
      }
#pragma unroll
      for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          acc0 = MFMA16(hw_bf2f(hv[i][j >> 1] & 0xffffu), bv[i][j], acc0);
          acc1 = MFMA16(hw_bf2f(hv[i][j >> 1] >> 16), bv[i][j + 1], acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
  }
  __syncthreads();

  // ---- phase B: waves 0..3; wave w does the softmax / loss / accuracy / dz2 math of rows 4g + w ----
  if (wave < 4) {
    const int row = brow;
    const bool row_ok = brow_ok;
    const float lab = (row_ok && cls_ok) ? labv : 0.f;
    float zsum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) zsum += red[w][wave][lane];       // fixed order: reproducible logits
    const float z = cls_ok ? zsum + biasv : NEG_INF;
    const float zmax = row16_max(z);
    const float e1 = cls_ok ? __expf(z - zmax) : 0.f;
    const float s1 = row16_sum(e1);
    const float y = e1 / s1;
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
    float loss_acc = (fr == 0 && row_ok) ? lrow_loss : 0.f;
    int correct_acc = (fr == 0 && row_ok && am_y == am_l) ? 1 : 0;
    const float dzv = (row_ok && cls_ok) ? dz : 0.f;
    dzs[4 * g + wave][fr] = dzv;
    if (row_ok && p.dz2) p.dz2[(size_t)row * 16 + fr] = dzv;
    loss_acc += __shfl_xor(loss_acc, 16);
    loss_acc += __shfl_xor(loss_acc, 32);
    correct_acc += __shfl_xor(correct_acc, 16);
    correct_acc += __shfl_xor(correct_acc, 32);
    if (lane == 0) {
      part_loss[wave] = loss_acc;
      part_cnt[wave] = correct_acc;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // rows are summed in a fixed order: the logged loss is reproducible
    if (p.loss_part) p.loss_part[blk] = (part_loss[0] + part_loss[1]) + (part_loss[2] + part_loss[3]);
    if (p.correct_part) p.correct_part[blk] = part_cnt[0] + part_cnt[1] + part_cnt[2] + part_cnt[3];
    if (p.post_part) p.post_part[blk] = 0;
  }
  if (!p.dz1_bf) return;

  // ---- phase C: dz1 = (dz2 @ W2^T) * (H > 0) over this wave's sixteenth of the hidden units, 16 per chunk.
  //      Result lane (fr, g) holds rows 4g..4g+3 of hidden unit chunk*16 + fr: the ReLU mask comes from H^T (the four
  //      rows are 8 contiguous bytes), the transposed output likewise; the row-major output is 2 bytes per row.
  float a4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = dzs[fr][4 * j + g];
  unsigned short* const bf = static_cast<unsigned short*>(p.dz1_bf);
  unsigned short* const bft = static_cast<unsigned short*>(p.dz1_bf_t);
  const int nc16 = (Hd + 15) >> 4;
  const int cpw = (nc16 + NW - 1) / NW;
  const int c_lo = wave * cpw, c_hi = min(nc16, c_lo + cpw);
  const int orow0 = r0 + 4 * g;
  const bool rows_ok = orow0 < M;                             // M is a multiple of 4: the four rows are inside or outside
  constexpr int CH2 = 8;
  for (int cb = c_lo; cb < c_hi; cb += CH2) {
    float cw[CH2][4];
    u32x2_t mk[CH2];
#pragma unroll
    for (int i = 0; i < CH2; ++i) {
      const int hcol = ((cb + i) << 4) + fr;
      const bool ok = cb + i < c_hi && hcol < Hd;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        cw[i][j] = bld(rw, (ok && 4 * j + g < C) ? ((unsigned)hcol * (unsigned)C + (unsigned)(4 * j + g)) * 4u : kOOB);
      mk[i] = hw_ld8(rht, (ok && rows_ok) ? ((unsigned)hcol * (unsigned)p.ldht + (unsigned)orow0) * 2u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < CH2; ++i) {
      const int hcol = ((cb + i) << 4) + fr;
      const bool ok = cb + i < c_hi && hcol < Hd && rows_ok;
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
      d0 = MFMA16(a4[0], cw[i][0], d0);
      d1 = MFMA16(a4[1], cw[i][1], d1);
      d0 = MFMA16(a4[2], cw[i][2], d0);
      d1 = MFMA16(a4[3], cw[i][3], d1);
      const float m4[4] = {hw_bf2f(mk[i][0] & 0xffffu), hw_bf2f(mk[i][0] >> 16), hw_bf2f(mk[i][1] & 0xffffu), hw_bf2f(mk[i][1] >> 16)};
      unsigned short hb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[r] = hw_f2bf(m4[r] > 0.f ? d0[r] + d1[r] : 0.f);
      if (ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bf[(size_t)(orow0 + r) * p.ld_bf + hcol] = hb[r];
        if (bft) {
          uint2 q;
          q.x = (unsigned)hb[0] | ((unsigned)hb[1] << 16);
          q.y = (unsigned)hb[2] | ((unsigned)hb[3] << 16);
          *reinterpret_cast<uint2*>(bft + (size_t)hcol * p.ld_bf_t + orow0) = q;
        }
      }
    }
  }
}

DTF_EXPORT int dtf_head_wide(const DtfHeadWideArgs* a, hipStream_t s) {
  if (!a || !a->H || !a->Ht || !a->W2 || !a->b2 || !a->labels) return DTF_ERR_ARG;
  if (a->M <= 0 || (a->M & 3) || a->Hd <= 0 || (a->Hd & 7) || a->C <= 0 || a->C > 16) return DTF_ERR_SHAPE;
  if (a->ldh < a->Hd || (a->ldh & 7) || a->ldht < a->M || (a->ldht & 3)) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->H) || (reinterpret_cast<uintptr_t>(a->Ht) & 7u)) return DTF_ERR_ALIGN;
  if (((uint64_t)(a->M - 1) * a->ldh + a->Hd) * 2u >= kOOB || ((uint64_t)(a->Hd - 1) * a->ldht + a->M) * 2u >= kOOB) return DTF_ERR_SHAPE;
  if ((uint64_t)a->Hd * a->C * 4u >= kOOB) return DTF_ERR_SHAPE;
  if (a->dz1_bf_t && !a->dz1_bf) return DTF_ERR_ARG;
  if (a->dz1_bf && a->ld_bf < a->Hd) return DTF_ERR_SHAPE;
  if (a->dz1_bf_t && (a->ld_bf_t < a->M || (a->ld_bf_t & 3) || (reinterpret_cast<uintptr_t>(a->dz1_bf_t) & 7u))) return DTF_ERR_ALIGN;
  hipLaunchKernelGGL(head_wide_kernel<16>, dim3(dtf_cdiv(a->M, 16)), dim3(16 * 64), 0, s, *a);
  return dtf_check_launch();
}

// Step tail for a large batch.  dtf_tail_body gives the indices of the next batch to 256 threads, each with a 64-bit
// division and two dependent loads per index: 16 serial rounds for 4096 rows, which made this one workgroup the longest
// of its launch.  Here every wave of the workgroup takes a share of the indices first (one 64-bit division per thread
// for the start of the batch, 32-bit arithmetic per index, the loads of a thread's indices all in flight), then waves
// 4.. end and the first 256 threads do the reduction and the counters exactly as dtf_tail_body does (a barrier does
// not wait for waves that have ended).
__device__ __forceinline__ void tail_wide(const DtfTailArgs& t, int tid, int nthreads) {
  const int ls = t.state[1];
  __syncthreads();                                    // every thread holds the counter before thread 0 advances it
  if (t.ridx_next) {
    const long long base = (long long)(ls + 1) * t.B;
    const long long epoch0 = base / t.n_train;
    const unsigned pos0 = (unsigned)(base - epoch0 * t.n_train);
    constexpr int U = 4;
    for (int i0 = tid; i0 < t.B; i0 += U * nthreads) {
      int row[U], prev[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * nthreads;
        const bool ok = i < t.B;
        const unsigned s = pos0 + (unsigned)(ok ? i : 0);           // < 2^32: both terms are below 2^31
        const unsigned wrap = s / (unsigned)t.n_train;
        const int pos = (int)(s - wrap * (unsigned)t.n_train);
        row[u] = t.perm ? t.perm[(size_t)((epoch0 + wrap) & 1) * t.n_train + pos] : pos;
        prev[u] = t.ridx_cur ? t.ridx_cur[ok ? i : 0] : row[u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * nthreads;
        if (i < t.B) {
          t.ridx_next[t.B + i] = prev[u];             // the batch just trained on becomes the accuracy-only rows
          t.ridx_next[i] = row[u];
        }
      }
    }
  }
  if (tid >= 256) return;
  DtfTailArgs r = t;
  r.ridx_next = nullptr;
  dtf_tail_body(r, tid, 256, false);
}

// dW2[Hd, C] = H^T[Hd, B] . dz2[B, 16],  db2 = colsum(dz2).  One workgroup = 16 hidden units; its 16 waves take
// contiguous sixteenths of the batch and their partial tiles are summed through LDS in wave order.
template <int NW>
__global__ __launch_bounds__(NW * 64) void out_wgrad_wide_kernel(DtfOutWgradWideArgs p) {
  const int ntile = (p.Hd + 15) >> 4;
  if ((int)blockIdx.x >= ntile) {
    if (p.tail.enable) tail_wide(p.tail, threadIdx.x, NW * 64);
    return;
  }
  __shared__ float part[NW][5][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int am = m0 + fr;
  const bool am_ok = am < p.Hd;
  const rsrc_t rht = make_rsrc(p.Ht, (unsigned)(((size_t)(p.Hd - 1) * p.ldht + p.B) * 2u));
  const rsrc_t ry = make_rsrc(p.dz2, (unsigned)p.B * 64u);
  const int nch = (p.B + 31) >> 5;
  const int cpw = (nch + NW - 1) / NW;
  const int c_lo = wave * cpw, c_hi = min(nch, c_lo + cpw);
  const unsigned a_row_off = (unsigned)am * (unsigned)p.ldht * 2u;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  constexpr int CH = 4;
  for (int cb = c_lo; cb < c_hi; cb += CH) {
    u32x4_t hv[CH];
    float bv[CH][8];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = ((cb + i) << 5) + 8 * g;
      const bool live = cb + i < c_hi && k < p.B;           // B is a multiple of 8
      hv[i] = hw_ld16(rht, (live && am_ok) ? a_row_off + (unsigned)k * 2u : kOOB);
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[i][j] = bld(ry, live ? ((unsigned)(k + j) * 16u + (unsigned)fr) * 4u : kOOB);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        bsum += bv[i][j] + bv[i][j + 1];
        acc0 = MFMA16(hw_bf2f(hv[i][j >> 1] & 0xffffu), bv[i][j], acc0);
        acc1 = MFMA16(hw_bf2f(hv[i][j >> 1] >> 16), bv[i][j + 1], acc1);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][r][lane] = acc0[r] + acc1[r];
  part[wave][4][lane] = bsum;
  __syncthreads();
  if (wave > 0) return;
  float out[4] = {0.f, 0.f, 0.f, 0.f};
  float bs = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] += part[w][r][lane];
    bs += part[w][4][lane];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + 4 * g + r;
    if (row < p.Hd && fr < p.C) p.dW[(size_t)row * p.ldw + fr] = out[r];
  }
  if (blockIdx.x == 0) {
    bs += __shfl_xor(bs, 16);
    bs += __shfl_xor(bs, 32);
    if (g == 0 && fr < p.C) p.db[fr] = bs;
  }
}

DTF_EXPORT int dtf_out_wgrad_wide(const DtfOutWgradWideArgs* a, hipStream_t s) {
  if (!a || !a->Ht || !a->dz2 || !a->dW || !a->db) return DTF_ERR_ARG;
  if (a->Hd <= 0 || a->B <= 0 || (a->B & 7) || a->C <= 0 || a->C > 16) return DTF_ERR_SHAPE;
  if (a->ldht < a->B || (a->ldht & 7) || a->ldw < a->C) return DTF_ERR_SHAPE;
  if (!dtf_aligned16(a->Ht)) return DTF_ERR_ALIGN;
  if (((uint64_t)(a->Hd - 1) * a->ldht + a->B) * 2u >= kOOB || (uint64_t)a->B * 64u >= kOOB) return DTF_ERR_SHAPE;
  if (a->tail.enable) {
    if (!a->tail.state) return DTF_ERR_ARG;
    if (a->tail.ridx_next && (a->tail.B <= 0 || a->tail.n_train <= 0 || a->tail.ridx_next == a->tail.ridx_cur)) return DTF_ERR_SHAPE;
    if (a->tail.nparts > 0 && (!a->tail.loss_part || !a->tail.correct_part || !a->tail.post_part)) return DTF_ERR_ARG;
  }
  hipLaunchKernelGGL(out_wgrad_wide_kernel<16>, dim3(dtf_cdiv(a->Hd, 16) + 1), dim3(16 * 64), 0, s, *a);
  return dtf_check_launch();
}
