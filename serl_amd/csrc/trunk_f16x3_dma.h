// Part of the split-fp16 trunk (trunk_f16x3.hip includes these in order; round 6 split the 2,600-line file by kernel family):
// conv_dma_f16x3_kernel: the LDS-DMA ring kernel (stages 1-3, stride-2 conv0s with the fused 1x1 projection) and its epilogues.
#pragma once
#include "trunk_f16x3_igemm.h"

namespace serl {

// ---------------------------------------------------------------------------------------------
// Epilogue of a 128 x (64*TN) tile held by 4 waves of 64 x (32*TN) (LDS-DMA kernels): combine the two accumulators, raw store or
// fused GroupNorm (+ residual) + ReLU + split8 store, statistics.
template <int TN, int PMODE>
__device__ __forceinline__ void dma_tile_epilogue(const ConvArgsB& ab, f32x16 (&acc)[2][TN], f32x16 (&accx)[2][TN], int m0, int n0,
                                                  int bn, int wm, int wn, int li, int lh) {
  constexpr int TM = 2, WROWS = 64, WCOLS = 32 * TN;
  const ConvArgs& a = ab.c;
  const int wrow0 = m0 + wm * WROWS;
  if (ab.wprio == 2) __builtin_amdgcn_s_setprio(1);
  float winv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) winv[tn] = ab.winv[n0 + wn * WCOLS + tn * 32 + li];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (acc[tm][tn][r] + accx[tm][tn][r] * kLoInv) * winv[tn];
  FusedResidual<TM, TN> fres;
  if ((PMODE == 0 || PMODE == 2) && ab.fz.mode) fused_load_residual<TM, TN>(ab, fres, wrow0, n0 + wn * WCOLS, li, lh);
  if (!ab.fz.mode) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wrow0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < a.M) {
          float* o = a.out + (size_t)m * a.Cout + n0 + wn * WCOLS + li;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) o[32 * tn] = acc[tm][tn][r];
        }
      }
  }
  if (PMODE != 3 && !(ab.fz.mode && !ab.fz.expected)) {   // (LOCAL fused mode keeps its statistics in the wave)
    const int gsize = a.Cout / kGnGroups;
    constexpr int ROWS = PMODE == 0 ? WROWS : (PMODE == 1 ? 32 : 16);
    constexpr int NSLOT = WROWS / ROWS;
#pragma unroll
    for (int slot = 0; slot < NSLOT; ++slot) {
      const int mrow = wrow0 + slot * ROWS;
      const bool valid = mrow < a.M;
      const int n = valid ? mrow / a.P : 0;
      double* stp = a.stats + (size_t)n * kGnGroups * 2;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + 8 * (r >> 2);
            if (row / ROWS == slot) {
              const float v = acc[tm][tn][r];
              s += v;
              q += v * v;
            }
          }
        stats_flush(s, q, stp, n0 + wn * WCOLS + tn * 32 + li, gsize, valid, ab.fz.mode != 0);   // (unfused: a later kernel reads them)
      }
    }
  }
  if (PMODE == 0 && ab.fz.mode && ab.fz.expected) {   // the launcher guarantees P % BM == 0: the whole tile lies in one image
    const int n_img = m0 / a.P;
    fused_arrive_and_wait(ab.fz.sync + n_img * a.tiles_n + bn, ab.fz.expected);
    fused_gn_store<TM, TN>(ab, acc, fres, n_img, wrow0, n0 + wn * WCOLS, li, lh);
  } else if (PMODE == 0 && ab.fz.mode) {              // LOCAL: this wave's block is one whole (image, group)
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float ps = 0.f, pq = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; ps += v; pq += v * v; }
      s += (double)ps; q += (double)pq;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
    const double mean = s * ab.fz.gn.inv_count, m2 = q * ab.fz.gn.inv_count;
    const float var = fmaxf((float)(m2 - mean * mean), 0.f);
    const float lmean = (float)mean, lrstd = rsqrtf(var + 1e-5f);
    fused_gn_store<TM, TN>(ab, acc, fres, wrow0 / a.P, wrow0, n0 + wn * WCOLS, li, lh, true, &lmean, &lrstd);
  } else if (PMODE == 2 && ab.fz.mode) {
    // LOCAL, stage 3 (round 6): images of 16 pixels, 128 channels per group.  The workgroup's 128 x 128 tile holds EIGHT whole images of ONE
    // group; a wave holds four of them (16-row slots) over half the group's channels, so the statistics of an (image, group) are this wave's
    // partial sums plus those of the wave next to it (wn ^ 1), exchanged through LDS -- no atomics, no arrival, no wait, and the two
    // elementwise launches that used to normalise this stage (gn_relu_split, block_out) are gone.
    __shared__ double s_x[4][4][2];   // [wave][slot][sum, sumsq]
    double s4[4], q4[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      float ps = 0.f, pq = 0.f;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 8 * (sl & 1); r < 8 * (sl & 1) + 8; ++r) { const float v = acc[sl >> 1][tn][r]; ps += v; pq += v * v; }
      s4[sl] = (double)ps; q4[sl] = (double)pq;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { s4[sl] += __shfl_xor(s4[sl], off); q4[sl] += __shfl_xor(q4[sl], off); }
    }
    const int wave = wm * 2 + wn;
    if (li == 0 && lh == 0) {
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) { s_x[wave][sl][0] = s4[sl]; s_x[wave][sl][1] = q4[sl]; }
    }
    __syncthreads();
    float lmean[4], lrstd[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      // (the two halves are added in column order on both waves: identical statistics for the whole group)
      const double ss = s_x[wm * 2][sl][0] + s_x[wm * 2 + 1][sl][0], qq = s_x[wm * 2][sl][1] + s_x[wm * 2 + 1][sl][1];
      const double mean = ss * ab.fz.gn.inv_count, m2 = qq * ab.fz.gn.inv_count;
      const float var = fmaxf((float)(m2 - mean * mean), 0.f);
      lmean[sl] = (float)mean; lrstd[sl] = rsqrtf(var + 1e-5f);
    }
    fused_gn_store<TM, TN, 4>(ab, acc, fres, wrow0 / a.P, wrow0, n0 + wn * WCOLS, li, lh, true, lmean, lrstd);
  }
}

// Raw store + GroupNorm statistics of the FUSED PROJECTION's tile (conv_dma_f16x3_kernel<.., PROJ = true>): the same tile geometry as the
// conv it rides on, so the same PMODE; never a fused GroupNorm epilogue (its consumer is conv1's residual operand, mode 3).
template <int TN, int PMODE>
__device__ __forceinline__ void dma_proj_epilogue(const ConvArgsB& ab, const ConvProjB& pj, f32x16 (&acc)[2][TN], f32x16 (&accx)[2][TN],
                                                  int m0, int n0, int wm, int wn, int li, int lh) {
  static_assert(PMODE != 3, "the fused projection takes its statistics in the kernel");
  constexpr int TM = 2, WROWS = 64, WCOLS = 32 * TN;
  const ConvArgs& a = ab.c;
  const int wrow0 = m0 + wm * WROWS;
  float winv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) winv[tn] = pj.winv[n0 + wn * WCOLS + tn * 32 + li];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (acc[tm][tn][r] + accx[tm][tn][r] * kLoInv) * winv[tn];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wrow0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < a.M) {
        float* o = pj.out + (size_t)m * a.Cout + n0 + wn * WCOLS + li;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) o[32 * tn] = acc[tm][tn][r];
      }
    }
  const int gsize = a.Cout / kGnGroups;
  constexpr int ROWS = PMODE == 0 ? WROWS : (PMODE == 1 ? 32 : 16);
  constexpr int NSLOT = WROWS / ROWS;
#pragma unroll
  for (int slot = 0; slot < NSLOT; ++slot) {
    const int mrow = wrow0 + slot * ROWS;
    const bool valid = mrow < a.M;
    const int n = valid ? mrow / a.P : 0;
    double* stp = pj.stats + (size_t)n * kGnGroups * 2;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tm * 32 + 8 * (r >> 2);
          if (row / ROWS == slot) {
            const float v = acc[tm][tn][r];
            s += v;
            q += v * v;
          }
        }
      stats_flush(s, q, stp, n0 + wn * WCOLS + tn * 32 + li, gsize, valid, false);   // (the projection's statistics are conv1's residual GroupNorm: a later kernel)
    }
  }
}

// LDS-DMA implicit GEMM (global_load_lds_dwordx4: HBM/L2 -> LDS without passing through registers).
// The register-staged kernel above serialises its phases -- measured on b2_conv1: MFMA-only 164 us, + LDS fragment
// reads 8, + ds_write staging 34, + global-load waits 57 = 263 us.  Here the K loop advances in 16-channel SLOTS
// (A: 128 rows x 64 B = [hi8 lo8 hi8 lo8] of the split8 layout, B: 64*TN rows x 64 B = [hi k0-7, hi k8-15, lo k0-7, lo k8-15])
// through a ring of four LDS positions:
//   * a slot's 16-byte LDS-DMA pieces are issued FOUR slots ahead (48 MFMAs = 1536 matrix-pipe cycles before use), one piece
//     behind every MFMA group (an LDS-DMA instruction costs ~60 issue cycles among MFMAs, several hundred when eight sit in
//     a row); the wait at the top of an iteration is a COUNTED vmcnt that leaves the two youngest slots in flight, one raw
//     s_barrier per slot;
//   * the fragments of slot c + 1 are read from LDS under the MFMAs of slot c (its ring position is refilled with slot c + 4
//     once every wave has passed the next barrier with lgkmcnt(0));
//   * a DMA piece is 64 lanes x 16 B written lane-linearly = 16 rows x 4 units, so the bank swizzle is applied on the SOURCE
//     side: lane l fetches unit (l & 3) ^ ((row >> 2) & 3) of row l >> 2, and the weights are pre-packed in piece order with
//     the swizzle baked in (pack_dma_order_kernel): conflict-free ds_read_b128;
//   * out-of-image taps fetch from a zero page (the DMA cannot zero-fill).
// Round 2's version (two LDS stages of 32 channels, a chunk's last piece issued right before the vmcnt(0) that waited for
// it) was 2-8 % slower per conv (b2_conv1 229 -> 210 us, b3_conv1 215 -> 193 us, same-call A/B; profiles/README.md).
// Timing-only ablation of this kernel on b2_conv1 (222 us on that box): MFMAs + barriers only 133 us (ideal at 2.4 GHz:
// 92 us -- the sustained clock under this load is ~1.7 GHz), + fragment reads 179, + DMA pieces (no reads) 184, DMA + reads
// without MFMAs 156, no barrier 226: reads and DMA cost ~50 us each ON TOP of the MFMA time wherever they sit in the
// instruction stream (pinning the order changed 214 -> 210 us), i.e. a shared-throughput / power cost, not exposed latency.
// Tile 128 x (64*TN) with 4 waves of 64 x (32*TN).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int TN, int PMODE, bool PROJ = false>
__global__ __launch_bounds__(256, 2) void conv_dma_f16x3_kernel(ConvArgsB ab, const uint8_t* zero_page, ConvProjB pj) {
  static_assert(!PROJ || PMODE != 3, "the fused projection takes its statistics in the kernel");
  constexpr int NS = 4;
  const ConvArgs& a = ab.c;
  constexpr int TM = 2, WROWS = 64, WCOLS = 32 * TN, BM = 128, BN = 2 * WCOLS;
  constexpr int A_BYTES = BM * 64, SLOT = A_BYTES + BN * 64;
  constexpr int A_PIECES = BM / 16 / 4;           // per wave per slot
  constexpr int B_PIECES = BN / 16 / 4;
  constexpr int PIECES = A_PIECES + B_PIECES;
  extern __shared__ __attribute__((aligned(16))) uint8_t smemb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  if (ab.wprio) __builtin_amdgcn_s_setprio(3);
  const int id = (ab.fz.mode && ab.fz.expected) ? fused_tile(ab.fz, gridDim.x) : xcd_remap((int)blockIdx.x, gridDim.x);
  const int bn = id % a.tiles_n, bm = id / a.tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int ntaps = a.KH * a.KW;
  unsigned rbase[A_PIECES], rmask[A_PIECES];
  const uint8_t* in_bytes = reinterpret_cast<const uint8_t*>(a.in);
#pragma unroll
  for (int q = 0; q < A_PIECES; ++q) {
    const int row = (q * 4 + wave) * 16 + (lane >> 2);
    const int u = (lane & 3) ^ ((row >> 2) & 3);
    const int m = m0 + row;
    rbase[q] = 0; rmask[q] = 0;
    if (m < a.M) {
      const int n = m / a.P, rem = m - n * a.P;
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      rbase[q] = (unsigned)((((long)(n * a.Hi + oy * a.stride) * a.Wi + ox * a.stride) * a.Cin) * 4 + u * 16);
      for (int t = 0; t < ntaps; ++t) {
        const int iy = oy * a.stride - a.pad + t / a.KW, ix = ox * a.stride - a.padw + t % a.KW;
        if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) rmask[q] |= 1u << t;
      }
    }
  }
  const int nslots = ntaps * (a.Cin >> 4);
  // The K loop below (SERL_RING_RUN) runs over one operand set: `run_nslots` slots of `run_kw`-wide kernel rows with the weight
  // pieces at wsrc.  Normally once; with a fused projection (PROJ) first over the projection's K = Cin (tap (0, 0) only).
  // (PROJ is a separate instantiation: the kernels without it compile to the same code as before the projection existed)
  constexpr bool with_proj = PROJ;
  int run_nslots = with_proj ? (a.Cin >> 4) : nslots, run_kw = with_proj ? 1 : a.KW;
  // weight pieces (ring-order copy: 4 KB per (64-row block, slot), swizzle baked in): this lane's 16 bytes of piece q
  const uint8_t* wsrc[B_PIECES];
#pragma unroll
  for (int q = 0; q < B_PIECES; ++q) {
    const int prow = (q * 4 + wave) * 16 + (lane >> 2);
    wsrc[q] = reinterpret_cast<const uint8_t*>(with_proj ? pj.wdma : ab.wdma) + (size_t)((n0 + prow) >> 6) * run_nslots * 4096 +
              ((prow & 63) << 6) + ((lane & 3) << 4);
  }
  int l_tap = 0, l_ky = 0, l_kx = 0, l_ci0 = 0, l_slot = 0;   // counters of the next slot to latch (strictly in order)
  const uint8_t* zp = zero_page + (lane & 3) * 16;
  int nx_tap = 0, nx_toff = 0, nx_k = 0, nx_ring = 0;
#define SERL_RING_PIECE(PI)                                                                                    \
  {                                                                                                            \
    uint8_t* st_ = smemb + nx_ring * SLOT;                                                                     \
    if ((PI) < A_PIECES) {                                                                                     \
      const int q = (PI) < A_PIECES ? (PI) : 0;                                                                \
      const bool ok = (rmask[q] >> nx_tap) & 1u;                                                               \
      const uint8_t* src = ok ? in_bytes + (size_t)rbase[q] + (long)nx_toff : zp;                              \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(st_ + (q * 4 + wave) * 1024), 16, 0, 0); \
    } else {                                                                                                   \
      const int q = (PI) >= A_PIECES ? (PI) - A_PIECES : 0;                                                    \
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc[q] + (size_t)nx_k * 4096),                           \
                                       (lds_void_t*)(st_ + A_BYTES + (q * 4 + wave) * 1024), 16, 0, 0);        \
    }                                                                                                          \
  }
  // past the last slot the counters stop: the last slot is fetched again into a free ring position (uniform loop, counted waits)
#define SERL_RING_NEXT(RING)                                                                                   \
  {                                                                                                            \
    nx_tap = l_tap; nx_k = l_slot; nx_ring = (RING);                                                           \
    nx_toff = (((l_ky - a.pad) * a.Wi + (l_kx - a.padw)) * a.Cin + l_ci0) * 4;                                 \
    if (l_slot + 1 < run_nslots) {                                                                             \
      ++l_slot;                                                                                                \
      l_ci0 += 16;                                                                                             \
      if (l_ci0 == a.Cin) { l_ci0 = 0; ++l_tap; if (++l_kx == run_kw) { l_kx = 0; ++l_ky; } }                  \
    }                                                                                                          \
  }
  f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accx[tm][tn][r] = 0.f; }
  const int li = lane & 31, lh = lane >> 5;
  int ahi_off[TM], alo_off[TM], bhi_off[TN], blo_off[TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = wm * WROWS + tm * 32 + li, sw = (row >> 2) & 3;
    ahi_off[tm] = row * 64 + (((2 * lh) ^ sw) << 4);
    alo_off[tm] = row * 64 + (((2 * lh + 1) ^ sw) << 4);
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int row = wn * WCOLS + tn * 32 + li, sw = (row >> 2) & 3;
    bhi_off[tn] = A_BYTES + row * 64 + ((lh ^ sw) << 4);
    blo_off[tn] = A_BYTES + row * 64 + (((2 + lh) ^ sw) << 4);
  }
  constexpr int GROUPS = TM * TN;
  f16x8 fa[2][2 * TM], fb[2][2 * TN];   // [register set][hi/lo per tile]
#define SERL_RING_READ(SET, ST)                                                                \
  {                                                                                            \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                        \
      fa[SET][2 * tm] = *reinterpret_cast<const f16x8*>((ST) + ahi_off[tm]);                   \
      fa[SET][2 * tm + 1] = *reinterpret_cast<const f16x8*>((ST) + alo_off[tm]);               \
    }                                                                                          \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                        \
      fb[SET][2 * tn] = *reinterpret_cast<const f16x8*>((ST) + bhi_off[tn]);                   \
      fb[SET][2 * tn + 1] = *reinterpret_cast<const f16x8*>((ST) + blo_off[tn]);               \
    }                                                                                          \
  }
  // iteration c: slot c is in register set CUR; slot c + 1 must have landed (slots c + 2, c + 3 may be in flight) and every
  // wave must be done reading slot c from LDS before its ring position is refilled with slot c + 4
#define SERL_RING_READ1(SET, ST, I)                                                            \
  {                                                                                            \
    if ((I) < 2 * TM) {                                                                        \
      const int tm_ = (I) >> 1;                                                                \
      fa[SET][I] = *reinterpret_cast<const f16x8*>((ST) + (((I) & 1) ? alo_off[tm_ < TM ? tm_ : 0] : ahi_off[tm_ < TM ? tm_ : 0])); \
    } else {                                                                                   \
      const int j_ = (I) - 2 * TM, tn_ = j_ >> 1;                                              \
      fb[SET][j_ < 2 * TN ? j_ : 0] = *reinterpret_cast<const f16x8*>((ST) + ((j_ & 1) ? blo_off[tn_ < TN ? tn_ : 0] : bhi_off[tn_ < TN ? tn_ : 0])); \
    }                                                                                          \
  }
  // iteration c: slot c is in register set CUR; slot c + 1 must have landed (slots c + 2, c + 3 may be in flight) and every
  // wave must be done reading slot c from LDS before its ring position is refilled with slot c + 4.  Per MFMA group the
  // instruction order is pinned with scheduling fences: cross MFMA 1, fragment reads of the next slot, hi*hi MFMA, one
  // LDS-DMA piece, cross MFMA 2 (hipcc otherwise sinks the reads behind the MFMAs that free their registers and issues
  // the DMA pieces back to back at the end of the iteration).
#define SERL_RING_ITER(C, CUR)                                                                 \
  {                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PIECES) : "memory");        \
    asm volatile("s_barrier" ::: "memory");                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    SERL_RING_NEXT((C) % NS);                                                                  \
    const uint8_t* stn = smemb + (((C) + 1) % NS) * SLOT;                                      \
    constexpr int NREAD = 2 * TM + 2 * TN, RPG = (NREAD + GROUPS - 1) / GROUPS;                \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                          \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                      \
        const int g = tm * TN + tn;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[CUR][2 * tm + 1], fb[CUR][2 * tn], accx[tm][tn], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < RPG; ++i) if (g * RPG + i < NREAD) SERL_RING_READ1(1 - (CUR), stn, g * RPG + i) \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[CUR][2 * tm], fb[CUR][2 * tn], acc[tm][tn], 0, 0, 0); \
        _Pragma("unroll") for (int pi = 0; pi < PIECES; ++pi)                                  \
          if (pi % GROUPS == g) SERL_RING_PIECE(pi)                         \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[CUR][2 * tm], fb[CUR][2 * tn + 1], accx[tm][tn], 0, 0, 0); \
      }                                                                                        \
  }
  // one pass over an operand set: prologue (NS slots in flight, slot 0's fragments into the first register set), the slot loop,
  // and the drain (the redundant last fetches must land before this LDS is reused or released)
#define SERL_RING_RUN()                                                                        \
  {                                                                                            \
    l_tap = 0; l_ky = 0; l_kx = 0; l_ci0 = 0; l_slot = 0;                                      \
    _Pragma("unroll") for (int p = 0; p < NS; ++p) {                                           \
      SERL_RING_NEXT(p);                                                                       \
      _Pragma("unroll") for (int pi = 0; pi < PIECES; ++pi) SERL_RING_PIECE(pi)                \
    }                                                                                          \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PIECES) : "memory");                   \
    asm volatile("s_barrier" ::: "memory");                                                    \
    SERL_RING_READ(0, smemb);                                                                  \
    for (int c = 0; c < run_nslots; c += 2) {   /* the slot count is even (Cin % 32 == 0) */   \
      SERL_RING_ITER(c, 0);                                                                    \
      SERL_RING_ITER(c + 1, 1);                                                                \
    }                                                                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
  }
  if constexpr (PROJ) {
    {   // the projection's tile first: K = Cin at conv0's tap (0, 0) -- launcher: stride 2, pad 0, 3x3
      SERL_RING_RUN();
      dma_proj_epilogue<TN, PMODE>(ab, pj, acc, accx, m0, n0, wm, wn, li, lh);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accx[tm][tn][r] = 0.f; }
      // every wave is done with the ring (its last fragment reads included) before conv0's prologue refills it, and the
      // projection's stores / statistics atomics have drained: the ring's counted vmcnt waits count DMA pieces only
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      run_nslots = nslots; run_kw = a.KW;
#pragma unroll
      for (int q = 0; q < B_PIECES; ++q) {
        const int prow = (q * 4 + wave) * 16 + (lane >> 2);
        wsrc[q] = reinterpret_cast<const uint8_t*>(ab.wdma) + (size_t)((n0 + prow) >> 6) * nslots * 4096 + ((prow & 63) << 6) + ((lane & 3) << 4);
      }
    }
  }
  SERL_RING_RUN();
#undef SERL_RING_RUN
#undef SERL_RING_ITER
#undef SERL_RING_READ
#undef SERL_RING_READ1
#undef SERL_RING_PIECE
#undef SERL_RING_NEXT
  dma_tile_epilogue<TN, PMODE>(ab, acc, accx, m0, n0, bn, wm, wn, li, lh);
}

}  // namespace serl
