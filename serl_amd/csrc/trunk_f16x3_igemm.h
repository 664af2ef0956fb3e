// Part of the split-fp16 trunk (trunk_f16x3.hip includes these in order; round 6 split the 2,600-line file by kernel family):
// conv_igemm_f16x3_kernel: the register-staged implicit-GEMM conv (small M, odd shapes, K-split of a rank's share).
#pragma once
#include "trunk_f16x3_common.h"

namespace serl {

// A operand = activations already in "split16" layout (written by the elementwise producers below):
// per 4 channels one 16-byte record {hi x4 fp16 | lo' x4 fp16}, i.e. the same footprint and addressing
// as the fp32 NHWC tensor.  The conv loader is then a pure 16-byte copy global -> LDS (zero VALU math).
// DEEP = 2 / 3: that many K chunks in flight in registers instead of one (0).  With the 64x64 tile (small M: one rank's share of a
// data-parallel batch) there is about one workgroup per CU and a chunk is only 6 MFMAs per wave, so the K loop runs
// at global-load latency (~1 us per chunk with one chunk in flight); the register budget of that tile allows more.
template <int WM, int WN, int TM, int TN, int PMODE, int DEEP = 0>
__global__ __launch_bounds__(256, 2) void conv_igemm_f16x3_kernel(ConvArgsB ab) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  const ConvArgs& a = ab.c;
  constexpr int WROWS = 32 * TM, WCOLS = 32 * TN;
  constexpr int BM = WROWS * WM, BN = WCOLS * WN;
  constexpr int AI = BM / 32;   // 16-byte A loads per thread per chunk
  constexpr int BI = BN / 32;   // 16-byte B loads per thread per chunk (hi and lo planes together)
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, STAGE = 2 * A_PLANE + 2 * B_PLANE;
  extern __shared__ __attribute__((aligned(16))) uint8_t smemb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int S = ab.ksplit > 1 ? ab.ksplit : 1;
  const int gid = xcd_remap(blockIdx.x, gridDim.x);   // (a tile's splits are neighbours: same XCD, shared operands in L2)
  const int id = gid / S, split = gid - id * S;
  const int bn = id % a.tiles_n, bm = id / a.tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  // per-thread im2col rows: element offset of the always-valid centre tap (pixel (oy*s, ox*s)) and a
  // bit mask of the taps that fall inside the image; out-of-image taps load the centre pixel and are
  // zeroed on the way to LDS, so the per-chunk address math is one select + one add per row.
  const int kq = tid & 7;
  const int ntaps = a.KH * a.KW;
  long rbase[AI];
  unsigned rmask[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + (tid >> 3) + 32 * i;
    rbase[i] = 0; rmask[i] = 0;
    if (m < a.M) {
      const int n = m / a.P, rem = m - n * a.P;
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      rbase[i] = ((long)(n * a.Hi + oy * a.stride) * a.Wi + ox * a.stride) * a.Cin + 4 * kq;
      for (int t = 0; t < ntaps; ++t) {
        const int iy = oy * a.stride - a.pad + t / a.KW, ix = ox * a.stride - a.padw + t % a.KW;
        if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) rmask[i] |= 1u << t;
      }
    }
  }
  const int nchunks_all = a.KH * a.KW * (a.Cin >> 5);
  const int cper = nchunks_all / S;                     // (the host picks S | nchunks_all)
  const int cb = split * cper;                          // first chunk of this workgroup
  const int nchunks = cb + cper;                        // one past its last chunk
  // (native vector types: HIP's uint4 struct copies lower to memcpy and keep the arrays out of registers)
  u32x4 ra[AI], rb[BI], ra2[DEEP >= 2 ? AI : 1], rb2[DEEP >= 2 ? BI : 1], ra3[DEEP >= 3 ? AI : 1], rb3[DEEP >= 3 ? BI : 1];
  unsigned okmask = 0, okmask2 = 0, okmask3 = 0;
  // chunk counters (chunks are visited strictly in order), started at chunk cb
  const int cpt = a.Cin >> 5;
  int l_tap = cb / cpt, l_ci0 = (cb - l_tap * cpt) << 5, l_ky = l_tap / a.KW, l_kx = l_tap - l_ky * a.KW;

#define SERL_LOAD_CHUNK_(CIDX, RA, RB, OK)                                                                                \
  {                                                                                                            \
    const int c_ = (CIDX);                                                                                     \
    const int tap = l_tap, ci0 = l_ci0;                                                                        \
    const int toff_ = ((l_ky - a.pad) * a.Wi + (l_kx - a.padw)) * a.Cin + ci0;                                 \
    /* advance the (tap, ky, kx, ci0) counters to the next chunk: no scalar divisions in the loop */          \
    if (c_ + 1 < nchunks) {                                                                                    \
      l_ci0 += 32;                                                                                             \
      if (l_ci0 == a.Cin) { l_ci0 = 0; ++l_tap; if (++l_kx == a.KW) { l_kx = 0; ++l_ky; } }                    \
    }                                                                                                          \
    OK = 0;                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                           \
      const bool ok = (rmask[i] >> tap) & 1u;                                                                  \
      OK |= (ok ? 1u : 0u) << i;                                                                               \
      RA[i] = *reinterpret_cast<const u32x4*>(a.in + rbase[i] + (ok ? toff_ : ci0));                           \
    }                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < BI; ++i) {                                                           \
      const int j_ = tid + 256 * i;                                                                            \
      const int plane_ = j_ / (BN * 4), r_ = (j_ / 4) % BN, s_ = j_ & 3;                                       \
      const int cc_ = min(c_, nchunks - 1);                                                                    \
      const uint16_t* wp_ = (plane_ ? ab.wlo : ab.whi) + (size_t)(n0 + r_) * ab.K + (cc_ << 5) + s_ * 8;                  \
      RB[i] = *reinterpret_cast<const u32x4*>(wp_);                                                            \
    }                                                                                                          \
  }
#define SERL_STORE_CHUNK_(BUF, RA, RB, OK)                                                                                  \
  {                                                                                                            \
    uint8_t* st_ = smemb + (BUF) * STAGE;                                                                      \
    _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                           \
      u32x4 v = RA[i];                                                                                         \
      if (!((OK >> i) & 1u)) v = (u32x4){0u, 0u, 0u, 0u};                                                      \
      const int row_ = (tid >> 3) + 32 * i;                                                                    \
      *reinterpret_cast<u32x4*>(st_ + (kq & 1) * A_PLANE + swz(row_, kq >> 1)) = v;  /* unit kq = plane kq&1 */ \
    }                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < BI; ++i) {                                                           \
      const int j_ = tid + 256 * i;                                                                            \
      const int plane_ = j_ / (BN * 4), r_ = (j_ / 4) % BN, s_ = j_ & 3;                                       \
      *reinterpret_cast<u32x4*>(st_ + 2 * A_PLANE + plane_ * B_PLANE + swz(r_, s_)) = RB[i]; \
    }                                                                                                          \
  }

  f32x16 acc[TM][TN], accx[TM][TN];  // hi*hi products / cross products (scaled by 2^11)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accx[tm][tn][r] = 0.f; }

#define SERL_LOAD_CHUNK(CIDX) SERL_LOAD_CHUNK_(CIDX, ra, rb, okmask)
#define SERL_STORE_CHUNK(BUF) SERL_STORE_CHUNK_(BUF, ra, rb, okmask)
#define SERL_COMPUTE_CHUNK(BUF)                                                                                \
  {                                                                                                            \
    const uint8_t* st = smemb + (BUF) * STAGE;                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
      f16x8 ahi[TM] = {}, alo[TM] = {}, bhi[TN] = {}, blo[TN] = {};                                            \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                                      \
        const int off = swz(wm * WROWS + tm * 32 + li, 2 * ks + lh);                                           \
        ahi[tm] = *reinterpret_cast<const f16x8*>(st + off);                                                   \
        alo[tm] = *reinterpret_cast<const f16x8*>(st + A_PLANE + off);                                         \
      }                                                                                                        \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                      \
        const int off = 2 * A_PLANE + swz(wn * WCOLS + tn * 32 + li, 2 * ks + lh);                             \
        bhi[tn] = *reinterpret_cast<const f16x8*>(st + off);                                                   \
        blo[tn] = *reinterpret_cast<const f16x8*>(st + B_PLANE + off);                                         \
      }                                                                                                        \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                                        \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                    \
          accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[tm], bhi[tn], accx[tm][tn], 0, 0, 0);      \
          accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[tm], blo[tn], accx[tm][tn], 0, 0, 0);      \
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[tm], bhi[tn], acc[tm][tn], 0, 0, 0);        \
        }                                                                                                      \
    }                                                                                                          \
  }
  const int li = lane & 31, lh = lane >> 5;
  if (!DEEP) {
    SERL_LOAD_CHUNK(cb);
    SERL_STORE_CHUNK(cb & 1);
    __syncthreads();
    for (int c = cb; c < nchunks; ++c) {
      const int buf = c & 1;
      SERL_LOAD_CHUNK(c + 1);  // (the last iteration re-reads its own chunk: the counters stop advancing)
      SERL_COMPUTE_CHUNK(buf);
      SERL_STORE_CHUNK(buf ^ 1);
      __syncthreads();
    }
  } else {
    // register set (c mod DEEP) holds chunk c+1 while chunk c is computed; its loads were issued DEEP iterations ago
    SERL_LOAD_CHUNK(cb);
    SERL_STORE_CHUNK(cb & 1);
    SERL_LOAD_CHUNK(cb + 1);
    SERL_LOAD_CHUNK_(cb + 2, ra2, rb2, okmask2);
    if (DEEP >= 3) SERL_LOAD_CHUNK_(cb + 3, ra3, rb3, okmask3);
    __syncthreads();
#define SERL_DEEP_STEP(C, RA, RB, OK)                    \
  {                                                      \
    SERL_COMPUTE_CHUNK((C) & 1);                         \
    SERL_STORE_CHUNK_(((C) + 1) & 1, RA, RB, OK);        \
    SERL_LOAD_CHUNK_((C) + 1 + DEEP, RA, RB, OK);        \
    __syncthreads();                                     \
  }
    for (int c = cb; c < nchunks; c += DEEP) {
      SERL_DEEP_STEP(c, ra, rb, okmask);
      if (c + 1 < nchunks) SERL_DEEP_STEP(c + 1, ra2, rb2, okmask2);
      if (DEEP >= 3 && c + 2 < nchunks) SERL_DEEP_STEP(c + 2, ra3, rb3, okmask3);
    }
#undef SERL_DEEP_STEP
  }
#undef SERL_COMPUTE_CHUNK
#undef SERL_LOAD_CHUNK_
#undef SERL_STORE_CHUNK_
#undef SERL_LOAD_CHUNK
#undef SERL_STORE_CHUNK

  const int wrow0 = m0 + wm * WROWS;
  float winv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) winv[tn] = ab.winv[n0 + wn * WCOLS + tn * 32 + li];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (acc[tm][tn][r] + accx[tm][tn][r] * kLoInv) * winv[tn];
  if (S > 1) {
    // K-split: publish this partial tile -- the accumulator registers as they are, four per 16-byte WRITE-THROUGH store (1 KB
    // per wave instruction, contiguous) -- drain, take a ticket; the last arriver re-reads all partials IN SPLIT ORDER with
    // L1-bypassing loads into the same registers and carries on below: the in-launch split-K recipe of
    // cdna_hip_programming.md (no fence, no spinning: nobody waits).  (4-byte partial stores, round 4's first build, are one
    // fabric write each and cost more than the K range saved: b3_conv1 85 -> 99 us.)
    constexpr int kSc1 = 16;
    constexpr int WAVE_FLOATS = TM * TN * 16 * 64;
    const size_t tile_floats = (size_t)4 * WAVE_FLOATS;
    // (ONE workgroup-uniform buffer descriptor for the tile's S partials; split, wave and lane go into the byte offset -- a
    //  descriptor whose base depends on the wave index lands in VGPRs and hipcc wraps every access in a waterfall loop)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ab.kslab + (size_t)id * S * tile_floats, 0, 0x7fffffff, 0x00020000);
    const int wl_off = (wave * WAVE_FLOATS + lane * 4) * 4;   // this lane's 16 bytes inside a quad block of its wave's region
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          typedef float f32x4s_t __attribute__((ext_vector_type(4)));
          const f32x4s_t vf = {acc[tm][tn][4 * q4], acc[tm][tn][4 * q4 + 1], acc[tm][tn][4 * q4 + 2], acc[tm][tn][4 * q4 + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vf), rs, split * (int)(tile_floats * 4) + wl_off + ((tm * TN + tn) * 4 + q4) * 1024, 0, kSc1);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smemb);   // (the operand LDS is idle: every wave passed the barrier above)
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(ab.kctr + id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const int last = old == S - 1;
      if (last) __hip_atomic_store(ab.kctr + id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
          for (int sp = 0; sp < S; ++sp)
            sum += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                       rs, sp * (int)(tile_floats * 4) + wl_off + ((tm * TN + tn) * 4 + q4) * 1024, 0, kSc1));
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[tm][tn][4 * q4 + j] = sum[j];
        }
  }
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
  if (PMODE != 3) {
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
        stats_flush(s, q, stp, n0 + wn * WCOLS + tn * 32 + li, gsize, valid, false);   // (never a fused launch: a later kernel reads them)
      }
    }
  }
}

}  // namespace serl
