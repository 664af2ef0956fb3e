// Part of the split-fp16 trunk (trunk_f16x3.hip includes these in order; round 6 split the 2,600-line file by kernel family):
// the row-slab kernels of the stride-1 3x3 convs of stage 0 and b1_conv1 (conv3x3_rowslab_f16x3_kernel: raw input;
// conv3x3_slabdma_f16x3_kernel: split8 input, everything by LDS-DMA) and their epilogues.
#pragma once
#include "trunk_f16x3_dma.h"

namespace serl {

// UNFUSED epilogue of a 256 x 64 output tile held by 4 waves of 64 x 64 (all rows in image n_img): combine the two accumulators,
// undo the weight scale, raw fp32 store, GroupNorm statistics (the elementwise pass then normalises).  The fused modes go through
// rowtile_epilogue_t below (the C-layout fused epilogue it replaced in round 5 -- 4-byte stores / residual loads, values traded between
// neighbour lanes by DPP -- and its switch SERL_EPI_T went with round 6; numbers in profiles/README.md).
__device__ __forceinline__ void rowtile_epilogue(const ConvArgsB& ab, f32x16 (&acc)[2][2], f32x16 (&accx)[2][2], int m0, int n0,
                                                 int n_img, int wave, int li, int lh) {
  const ConvArgs& a = ab.c;
  constexpr int TM = 2, TN = 2, WROWS = 64;
  const int wrow0 = m0 + wave * WROWS;
  float winv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) winv[tn] = ab.winv[n0 + tn * 32 + li];
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
      float* o = a.out + (size_t)m * a.Cout + n0 + li;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) o[32 * tn] = acc[tm][tn][r];
    }
  const int gsize = a.Cout / kGnGroups;
  double* stp = a.stats + (size_t)n_img * kGnGroups * 2;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[tm][tn][r];
        s += v;
        q += v * v;
      }
    stats_flush(s, q, stp, n0 + tn * 32 + li, gsize, true, false);   // (unfused: a later kernel reads them)
  }
}

// The fused GroupNorm (+ residual) + ReLU + split8 epilogue of a 256 x 64 row tile, ROW-MAJOR (round 5).  A store straight
// from the MFMA C layout means a lane owns ONE channel of 16 rows per 32 x 32 tile, i.e. 64 four-byte stores and (with a
// residual) 64 four-byte loads per lane, half of the values traded with the neighbour lane by DPP: 2.7 TB/s on the store-only
// epilogue of b0_conv0, issue-bound.  Here every wave writes its 64 x 64 accumulator tile to the (now idle) operand LDS once,
// 16 KB per wave, and reads it back with a lane owning EIGHT consecutive channels of a row: the residual arrives as two 16-byte
// loads, the split8 record (16 bytes of hi halves + 16 bytes of lo' halves) leaves as two 16-byte stores, eight lanes cover a
// row's 256 contiguous bytes -- 16 + 16 wide memory instructions per lane instead of 64 + 64 narrow ones, no lane exchange.
// Same arithmetic per element as fused_gn_store.  Statistics, arrival and wait are unchanged (taken from the registers first).
__device__ __forceinline__ void rowtile_epilogue_t(const ConvArgsB& ab, f32x16 (&acc)[2][2], f32x16 (&accx)[2][2], int m0, int n0,
                                                   int n_img, int wave, int lane, int sync_idx, uint8_t* lds) {
  const ConvArgs& a = ab.c;
  const FuseArgs& fz = ab.fz;
  constexpr int TM = 2, TN = 2, WROWS = 64;
  const int li = lane & 31, lh = lane >> 5;
  const int wrow0 = m0 + wave * WROWS;
  if (ab.wprio == 2) __builtin_amdgcn_s_setprio(1);   // (the epilogue yields to the main loop of the CU's other workgroup)
  float winv[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) winv[tn] = ab.winv[n0 + tn * 32 + li];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = (acc[tm][tn][r] + accx[tm][tn][r] * kLoInv) * winv[tn];
  {
    // Statistics: the four waves of the tile cover the SAME 64 channels of the SAME image, so their partial sums are added in the
    // workgroup first (through LDS, in wave order, in fp64) and ONE wave issues the returning atomics: 8 per tile instead of 32.  The
    // atomics are contended at the memory side -- 4 tiles x 4 waves x 8 on the eight (sum, sumsq) words of an image -- and the timing
    // ablation without them ran the stage-0 convs 22-37 us shorter; issuing a wave's four at once instead of two and two made the
    // kernels SLOWER (profiles/README.md round 6), fewer of them is what helps.  Ordering as before: the results are consumed before
    // the barrier in front of the arrival (fused_arrive_and_wait).
    __shared__ float s_part[4][8];   // [wave][(tn * 2 + 16-channel segment) * 2 + {sum, sumsq}]
    const int gsize = a.Cout / kGnGroups;
    double* stp = a.stats + (size_t)n_img * kGnGroups * 2;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; s += v; q += v * v; }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if ((lane & 15) == 0 && lane < 32) {
        s_part[wave][(tn * 2 + (lane >> 4)) * 2] = s;
        s_part[wave][(tn * 2 + (lane >> 4)) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 4) {   // lane = tn * 2 + segment
      const double ss = (((double)s_part[0][2 * lane] + (double)s_part[1][2 * lane]) + (double)s_part[2][2 * lane]) + (double)s_part[3][2 * lane];
      const double qq = (((double)s_part[0][2 * lane + 1] + (double)s_part[1][2 * lane + 1]) + (double)s_part[2][2 * lane + 1]) + (double)s_part[3][2 * lane + 1];
      const int g = (n0 + 16 * lane) / gsize;
      const double o0 = __hip_atomic_fetch_add(&stp[2 * g], ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const double o1 = __hip_atomic_fetch_add(&stp[2 * g + 1], qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("" ::"v"(o0), "v"(o1));
    }
  }
  // the wave's tile -> LDS [row][64 floats] (every wave passed the main loop's last barrier: the operand buffers are idle)
  float* tile = reinterpret_cast<float*>(lds) + wave * (64 * 64);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + tn * 32 + li] = acc[tm][tn][r];
  // this lane's eight channels and its rows (8 lanes per row, 8 rows per pass); residual of the first passes requested before the wait
  const int g8 = lane & 7, rsub = lane >> 3, c0 = n0 + 8 * g8;
  const size_t rowb = (size_t)a.Cout * 4;
  const uint8_t* res_base = fz.mode == 2 ? fz.res_split + (size_t)wrow0 * rowb + c0 * 4
                                         : reinterpret_cast<const uint8_t*>(fz.res_raw) + (size_t)wrow0 * rowb + c0 * 4;
  u32x4 rres[8][2];
  if (fz.mode >= 2) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint8_t* q = res_base + (size_t)(8 * p + rsub) * rowb;
      rres[p][0] = *reinterpret_cast<const u32x4*>(q);
      rres[p][1] = *reinterpret_cast<const u32x4*>(q + 16);
    }
  }
  fused_arrive_and_wait(fz.sync + sync_idx, fz.expected);
  float sc[8], sh[8], rs[8], rh[8];
  {
    const double* st = fz.gn.stats + ((size_t)n_img * kGnGroups + c0 / fz.gn.gsize) * 2;   // (8 consecutive channels: one group)
    const double s0 = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double s1 = __hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double mean = s0 * fz.gn.inv_count, m2 = s1 * fz.gn.inv_count;
    const float var = fmaxf((float)(m2 - mean * mean), 0.f);
    const float rstd = rsqrtf(var + 1e-5f), mf = (float)mean;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = fz.gn.gamma[c0 + j] * rstd; sh[j] = fz.gn.beta[c0 + j] - mf * sc[j]; rs[j] = 0.f; rh[j] = 0.f; }
    if (fz.mode >= 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j) gn_coef1<false>(fz.res_gn, n_img, c0 + j, rs[j], rh[j]);
    }
  }
  uint8_t* out_base = fz.out_split + (size_t)wrow0 * rowb + c0 * 4;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 8 * p + rsub;
    const float4 t0 = *reinterpret_cast<const float4*>(tile + row * 64 + 8 * g8);
    const float4 t1 = *reinterpret_cast<const float4*>(tile + row * 64 + 8 * g8 + 4);
    float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
    if (fz.mode == 2) {          // residual in split8 form: 8 hi halves | 8 lo' halves
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (through scalars: __builtin_bit_cast applied to a vector-ELEMENT lvalue reads element 0 with this compiler)
        const uint32_t wh = rres[p][0][j], wl = rres[p][1][j];
        const h16x2 hh = __builtin_bit_cast(h16x2, wh), ll = __builtin_bit_cast(h16x2, wl);
        v[2 * j] = ((float)hh[0] + (float)ll[0] * kLoInv) + v[2 * j];
        v[2 * j + 1] = ((float)hh[1] + (float)ll[1] * kLoInv) + v[2 * j + 1];
      }
    } else if (fz.mode >= 3) {   // raw fp32 residual: GroupNorm of the projection (3) or relu(GroupNorm) of the block input (4)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t wx = rres[p][j >> 2][j & 3];
        const float x = __builtin_bit_cast(float, wx);
        const float y = x * rs[j] + rh[j];
        v[j] = (fz.mode == 4 ? fmaxf(y, 0.f) : y) + v[j];
      }
    }
    u32x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a0 = fmaxf(v[2 * j], 0.f), a1 = fmaxf(v[2 * j + 1], 0.f);
      const h16x2 hp = __builtin_amdgcn_cvt_pkrtz(a0, a1);
      const f32x2 rem = {(a0 - (float)hp[0]) * kLoScale, (a1 - (float)hp[1]) * kLoScale};
      const f16x2 lp = __builtin_convertvector(rem, f16x2);
      hi[j] = __builtin_bit_cast(uint32_t, hp);
      lo[j] = __builtin_bit_cast(uint32_t, lp);
    }
    uint8_t* o = out_base + (size_t)row * rowb;
    *reinterpret_cast<u32x4*>(o) = hi;
    *reinterpret_cast<u32x4*>(o + 16) = lo;
  }
}

// ---------------------------------------------------------------------------------------------
// Row-slab kernel for the stride-1 3x3 convs of stage 0 and b1_conv1 (the largest M and the smallest N, where an im2col
// loader's 9x re-read of every input pixel through L2 -> LDS is the bound): 256 x 64 output tile = 256 / Wo whole output
// rows of one image.  (Its round-2 predecessor, the row-patch kernel -- one kernel row per chunk, activations crossing
// L2 -> LDS 3x -- moved 1.43 GB per launch at 4.8 TB/s with the matrix pipe 27 % busy and zero LDS conflicts: traffic-
// bound; removed in round 3, numbers in profiles/README.md.)  The K loop is channel-major: for each group of 16 input channels the workgroup stages the (TR + 2) x (Wo + 2)
// input pixels ONCE (a slab: 22 KB) and serves all NINE taps from it (ky shifts the row, kx the pixel); only the 3 taps'
// weights (13 KB) are streamed per (channel group, ky) sub-chunk.  The next slab is fetched in three parts under the three
// sub-chunks of the current one, so a thread stages 2 activation units + 3 weight units per sub-chunk (20 registers
// instead of 32).  Activation traffic 835 -> 357 MB per launch (the operand crosses L2 -> LDS 1.25x instead of 3x).
// ---------------------------------------------------------------------------------------------
constexpr int kRowslabPix = 340;   // (8 + 2) x 34 (Wo = 32); (16 + 2) x 18 = 324 (Wo = 16)
constexpr int kRowslabLds = 2 * 4 * (kRowslabPix * 16 + 32) + 2 * 3 * 2 * 2 * (64 * 16 + 64);
// The input is the RAW fp32 tensor of the producing layer (conv_init's completed pooling output): GroupNorm + ReLU + the hi / lo'
// split are applied while a slab is staged (a.in_gn: per-channel scale / shift of this tile's image, held in LDS) -- the
// elementwise pass that would materialise the split8 tensor (read 268 MB + write 268 MB per trunk pass) is gone.  A thread stages
// one (pixel, k-half) = 8 channels per slab part: two 16-byte fp32 loads in, one hi and one lo' unit out.  A split8 input goes to
// conv3x3_slabdma_f16x3_kernel below (everything by LDS-DMA); the register-staged split8 form of this kernel and its switch
// SERL_SLAB_DMA went with round 6 (numbers in profiles/README.md).
// The WEIGHTS of a sub-chunk arrive by LDS-DMA in the LDS-DMA kernels' piece order (4 KB per tap, swizzled [cout][64 B]: see
// conv3x3_slabdma_f16x3_kernel) one sub-chunk ahead; the activations keep the register path (GroupNorm + ReLU + split on the way).
// Always a FUSED launch (a raw input exists only when the fused pass hands it over): the row-major epilogue rowtile_epilogue_t.
__global__ __launch_bounds__(256, 2) void conv3x3_rowslab_f16x3_kernel(ConvArgsB ab) {
  const ConvArgs& a = ab.c;
  constexpr int TM = 2, TN = 2, WROWS = 64, BM = 256, BN = 64;
  // LDS image: 16-byte units (8 fp16 = one MFMA k-half of one plane) laid out so that the 32 lanes of an MFMA fragment read
  // (consecutive pixels, same plane and k-half) touch CONSECUTIVE units -- ds_read_b128 serves 16-lane groups over a 256-byte
  // bank row, and [pixel][32 B] rows would put lanes l and l+8 of a group on the same banks (PMC: 45 % of the LDS cycles were
  // bank conflicts with that layout).  Activations: 4 regions (plane, k-half) of [pixel][16 B], region q = plane + 2 * k-half at
  // q * A_REGION; weights: three taps of 4 KB per buffer in the DMA piece order.  Region strides are padded so that the 8 lanes
  // of a ds_write_b128 group (2 pixels x 4 regions) cover all 32 write banks.
  constexpr int A_REGION = kRowslabPix * 16 + 32, A_BYTES = 4 * A_REGION;
  constexpr int B_HALF = BN * 16 + 64, B_PLANE = 2 * B_HALF, B_TAP = 2 * B_PLANE, B_BYTES = 3 * B_TAP;
  static_assert(2 * A_BYTES + 2 * B_BYTES == kRowslabLds, "LDS size of the launch");
  static_assert(3 * 4096 <= B_BYTES, "three DMA pieces per weight buffer");
  constexpr int AJ = 2;   // activation units per thread per sub-chunk: 3 x 2 x 256 = 1536 >= 340 x 4
  static_assert(3 * AJ * 256 >= kRowslabPix * 4, "a slab is fetched in three parts");
  extern __shared__ __attribute__((aligned(16))) uint8_t smemb[];
  uint8_t* const smA = smemb;
  uint8_t* const smB = smemb + 2 * A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ANTI-PHASE START.  A tile is a matrix-bound main loop followed by an HBM-bound fused epilogue (residual read + split8 write:
  // 114 of b0_conv1's 351 us), all 512 resident workgroups start together and every tile takes the same time, so the chip
  // alternates between "all MFMA, HBM idle" and "all HBM at 4.7 TB/s, matrix pipe idle".  Delaying every CU's second
  // workgroup by about half a tile BEFORE it draws its ticket shifts half of the tiles by half a period for the rest of the
  // launch (a finished workgroup's slot is refilled at once, tickets are handed out in start order, so the tiles of one image
  // still start together and wait for nobody longer than before).
  if (ab.wprio) __builtin_amdgcn_s_setprio(3);
  if (ab.stagger > 0 && blockIdx.x >= 256u && blockIdx.x < 512u)
    for (int i = 0; i < ab.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  const int id = ab.fz.mode ? fused_tile(ab.fz, gridDim.x) : xcd_remap((int)blockIdx.x, gridDim.x);
  const int bn = id % a.tiles_n, bm = id / a.tiles_n;   // 64-channel column tiles of one row tile are neighbours (shared slab in L2)
  const int m0 = bm * BM, n0 = bn * BN;
  const int n_img = m0 / a.P, oy0 = (m0 - n_img * a.P) / a.Wo;
  const int pw = a.Wo + 2, npix = (BM / a.Wo + 2) * pw;
  const int c16n = a.Cin >> 4, nchunks = 3 * c16n;
  int rbase[3][AJ];        // element offset of unit (part, j) of a slab at channel group 0 (clamped into the image)
  unsigned okbits = 0;     // bit part*AJ + j: the unit's pixel lies inside the image (else it is stored as zeros)
  __shared__ float s_gn[2][128];   // GroupNorm scale / shift per input channel of this tile's image
  if (tid < a.Cin) gn_coef1<false>(a.in_gn, n_img, tid, s_gn[0][tid], s_gn[1][tid]);
#pragma unroll
  for (int part = 0; part < 3; ++part) {   // (pixel, k-half) = part * 256 + tid; both staging registers belong to it
    const int v = part * 256 + tid, pix = v >> 1, kh = v & 1;
    const int sy = pix / pw, sx = pix - sy * pw;
    const int iy = oy0 - 1 + sy, ix = sx - 1;
    const int iyc = min(max(iy, 0), a.Hi - 1), ixc = min(max(ix, 0), a.Wi - 1);
    rbase[part][0] = ((n_img * a.Hi + iyc) * a.Wi + ixc) * a.Cin + 8 * kh;
    rbase[part][1] = rbase[part][0] + 4;
    if (pix < npix && iy == iyc && ix == ixc) okbits |= 3u << (part * AJ);
  }
  __syncthreads();
  u32x4 ra[AJ];

// fetch into registers: part PART of the slab of channel group CG (activations); by DMA: the 3 taps of (channel group BG, row BKY)
#define SERL_RS_LOAD_A(RA, PART, CG)                                                               \
  _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                   \
    RA[j] = *reinterpret_cast<const u32x4*>(a.in + rbase[PART][j] + ((CG) << 4));
#define SERL_RS_DMA_B(BG, BKY, BBUF)                                                               \
  _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                 \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(wdsrc + ((size_t)(((BKY) * 3 + kx) * c16n + (BG)) << 12)), \
                                     (lds_void_t*)(smB + (BBUF) * B_BYTES + kx * 4096 + (tid >> 6) * 1024), 16, 0, 0);
#define SERL_RS_STORE_A(RA, PART, ABUF, CGN)                                                       \
  {                                                                                                \
    const int v_ = (PART) * 256 + tid, kh_ = v_ & 1, cb_ = ((CGN) << 4) + 8 * kh_;                 \
    u32x4 hi_ = {0u, 0u, 0u, 0u}, lo_ = {0u, 0u, 0u, 0u};                                          \
    if ((okbits >> ((PART) * AJ)) & 1u) {                                                          \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
        const float4 x_ = __builtin_bit_cast(float4, RA[j]);                                       \
        const float4 sc_ = *reinterpret_cast<const float4*>(&s_gn[0][cb_ + 4 * j]);                \
        const float4 sh_ = *reinterpret_cast<const float4*>(&s_gn[1][cb_ + 4 * j]);                \
        const float4 y_ = make_float4(fmaxf(x_.x * sc_.x + sh_.x, 0.f), fmaxf(x_.y * sc_.y + sh_.y, 0.f), \
                                      fmaxf(x_.z * sc_.z + sh_.z, 0.f), fmaxf(x_.w * sc_.w + sh_.w, 0.f)); \
        uint2 h2_, l2_;                                                                            \
        split4(y_, h2_, l2_);                                                                      \
        hi_[2 * j] = h2_.x; hi_[2 * j + 1] = h2_.y; lo_[2 * j] = l2_.x; lo_[2 * j + 1] = l2_.y;    \
      }                                                                                            \
    }                                                                                              \
    if (v_ < kRowslabPix * 2) {                                                                    \
      uint8_t* d_ = smA + (ABUF) * A_BYTES + (2 * kh_) * A_REGION + (v_ >> 1) * 16;                \
      *reinterpret_cast<u32x4*>(d_) = hi_;                                                         \
      *reinterpret_cast<u32x4*>(d_ + A_REGION) = lo_;                                              \
    }                                                                                              \
  }
// what is fetched while sub-chunk (CG, KY) computes: the weights of the NEXT sub-chunk and part KY of the NEXT slab (the
// last slab re-fetches itself: harmless, keeps the loop uniform) -- and where it goes when that sub-chunk is done
#define SERL_RS_LOADS(CG, KY, RA)                                                                  \
  {                                                                                                \
    const int ncg_ = (KY) == 2 ? (CG) + 1 : (CG), nky_ = (KY) == 2 ? 0 : (KY) + 1;                 \
    const int ncgc_ = min(ncg_, c16n - 1), sn_ = min((CG) + 1, c16n - 1);                          \
    SERL_RS_DMA_B(ncgc_, nky_, ((CG) * 3 + (KY) + 1) & 1)                                          \
    if ((KY) == 0) { SERL_RS_LOAD_A(RA, 0, sn_); } else if ((KY) == 1) { SERL_RS_LOAD_A(RA, 1, sn_); } else { SERL_RS_LOAD_A(RA, 2, sn_); } \
  }
#define SERL_RS_STORES(CG, KY, RA)                                                                 \
  {                                                                                                \
    const int sn2_ = min((CG) + 1, c16n - 1);                                                      \
    if ((KY) == 0) { SERL_RS_STORE_A(RA, 0, ((CG) + 1) & 1, sn2_); } else if ((KY) == 1) { SERL_RS_STORE_A(RA, 1, ((CG) + 1) & 1, sn2_); } \
    else { SERL_RS_STORE_A(RA, 2, ((CG) + 1) & 1, sn2_); }                                         \
  }
#define SERL_RS_COMPUTE(C, CG, KY)                                                                 \
  {                                                                                                \
    const uint8_t* sa = smA + ((CG) & 1) * A_BYTES + (KY) * pw * 16;                               \
    const uint8_t* sb = smB + ((C) & 1) * B_BYTES;                                                 \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                             \
      f16x8 ahi[TM], alo[TM], bhi[TN], blo[TN];                                                    \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                          \
        ahi[tm] = *reinterpret_cast<const f16x8*>(sa + arow[tm] + kx * 16);                        \
        alo[tm] = *reinterpret_cast<const f16x8*>(sa + A_REGION + arow[tm] + kx * 16);             \
      }                                                                                            \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                          \
        bhi[tn] = *reinterpret_cast<const f16x8*>(sb + kx * 4096 + wd_bhi[tn]);                    \
        blo[tn] = *reinterpret_cast<const f16x8*>(sb + kx * 4096 + wd_blo[tn]);                    \
      }                                                                                            \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                            \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                        \
          accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[tm], bhi[tn], accx[tm][tn], 0, 0, 0); \
          accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[tm], blo[tn], accx[tm][tn], 0, 0, 0); \
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[tm], bhi[tn], acc[tm][tn], 0, 0, 0); \
        }                                                                                          \
    }                                                                                              \
  }

  f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accx[tm][tn][r] = 0.f; }

  const int li = lane & 31, lh = lane >> 5;
  int arow[TM];  // LDS byte offset of this lane's pixel at (ky, kx) = (0, 0) for each 32-row MFMA tile
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = wave * WROWS + tm * 32 + li;
    const int y = r / a.Wo, x = r - y * a.Wo;
    arow[tm] = 2 * lh * A_REGION + (y * pw + x) * 16;
  }
  int wd_bhi[TN], wd_blo[TN];   // swizzled [cout][64 B] image of a tap's weights
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int row = tn * 32 + li, sw = (row >> 2) & 3;
    wd_bhi[tn] = row * 64 + ((lh ^ sw) << 4);
    wd_blo[tn] = row * 64 + (((2 + lh) ^ sw) << 4);
  }
  const uint8_t* wdsrc = reinterpret_cast<const uint8_t*>(ab.wdma) + (size_t)(n0 >> 6) * (9 * c16n) * 4096 + (tid >> 6) * 1024 + lane * 16;
  // prologue: slab 0 (three parts) and the weights of sub-chunk 0
  SERL_RS_DMA_B(0, 0, 0)
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    SERL_RS_LOAD_A(ra, part, 0);
    SERL_RS_STORE_A(ra, part, 0, 0);
  }
  __syncthreads();
  int cg = 0, ky = 0;   // channel group and kernel row of sub-chunk c
  for (int c = 0; c < nchunks; ++c) {
    SERL_RS_LOADS(cg, ky, ra);
    SERL_RS_COMPUTE(c, cg, ky);
    // the other weight buffer was last read in sub-chunk c - 1, the other slab buffer during the previous channel group
    // (fetching two sub-chunks ahead with a second staging register set was measured neutral in round 2: removed)
    SERL_RS_STORES(cg, ky, ra);
    __syncthreads();
    if (++ky == 3) { ky = 0; ++cg; }
  }
#undef SERL_RS_LOAD_A
#undef SERL_RS_DMA_B
#undef SERL_RS_STORE_A
#undef SERL_RS_LOADS
#undef SERL_RS_STORES
#undef SERL_RS_COMPUTE
  rowtile_epilogue_t(ab, acc, accx, m0, n0, n_img, wave, lane, n_img * a.tiles_n + bn, smemb);
}

// ---------------------------------------------------------------------------------------------
// Row-slab kernel with LDS-DMA staging (round 5): the tile geometry, the K order (16-channel
// groups, per group three sub-chunks = kernel rows, three taps each) and the epilogue of conv3x3_rowslab_f16x3_kernel, but the
// operands go HBM / L2 -> LDS by global_load_lds_dwordx4 as in the ring kernel -- no staging registers, no ds_write pass, no
// per-unit zeroing selects.  Input must be split8 (no RAWIN: GroupNorm cannot be applied by a DMA).
//   * slab image: [pixel][64 B] = the four 16-byte units of a 16-channel group (hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15), slot
//     s of pixel (sy, sx) holding unit s ^ ((sx >> 2) & 3).  A swizzle by the COLUMN only: a tap shifts (sy, sx) by (ky, kx), so a
//     lane's offsets for the three kernel rows differ by a constant and only depend on kx.  Conflict-free ds_read_b128 for every
//     tap when a slab row starts on a multiple of four pixels or the map is 32 wide: pitch 34 (Wo = 32), 20 (Wo = 16, two pad
//     pixels per row) -- checked by enumeration over the hardware's 16-lane groups (profiles/README.md round 5).
//   * a DMA piece = 16 pixels x 4 slots, lane l fetching unit (l & 3) ^ ((sx >> 2) & 3) of pixel 16 p + (l >> 2) (source-side
//     swizzle); pixels outside the image / the slab fetch a zero page.  23 pieces per slab, wave w takes pieces w, w + 4, ...;
//     two per sub-chunk, into the slab buffer of the NEXT channel group;
//   * weights: the LDS-DMA kernel's piece order (pack_dma_order_kernel: 4 KB per (64 couts, 16-wide K slot), swizzle baked in),
//     slot (tap, cg) = tap * Cin / 16 + cg; a sub-chunk's three taps = 12 pieces, three per wave, one sub-chunk ahead.
// One barrier per sub-chunk (36 MFMAs per wave), every DMA waited for with vmcnt(0) a whole sub-chunk after its issue.
// ---------------------------------------------------------------------------------------------
constexpr int kSdPieces = 23;                          // 368 pixels >= 18 x 20 (Wo = 16) and >= 10 x 34 (Wo = 32)
constexpr int kSdSlab = kSdPieces * 1024, kSdW = 3 * 4096;
constexpr int kSlabDmaLds = 2 * kSdSlab + 2 * kSdW;    // 71,680 B: two workgroups per CU leave 16 KB for a chain GEMM workgroup

template <bool EPT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_slabdma_f16x3_kernel(ConvArgsB ab, const uint8_t* zero_page) {
  const ConvArgs& a = ab.c;
  constexpr int TM = 2, TN = 2, WROWS = 64, BM = 256, BN = 64;
  extern __shared__ __attribute__((aligned(16))) uint8_t smemb[];
  uint8_t* const smS = smemb;
  uint8_t* const smW = smemb + 2 * kSdSlab;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (ab.wprio) __builtin_amdgcn_s_setprio(3);
  if (ab.stagger > 0 && blockIdx.x >= 256u && blockIdx.x < 512u)   // anti-phase start, see conv3x3_rowslab_f16x3_kernel
    for (int i = 0; i < ab.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  const int id = ab.fz.mode ? fused_tile(ab.fz, gridDim.x) : xcd_remap((int)blockIdx.x, gridDim.x);
  const int bn = id % a.tiles_n, bm = id / a.tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int n_img = m0 / a.P, oy0 = (m0 - n_img * a.P) / a.Wo;
  const int pw = a.Wo == 32 ? 34 : 20;
  const int srows = BM / a.Wo + 2;
  const int c16n = a.Cin >> 4, nchunks = 3 * c16n, nslots = 9 * c16n;
  const uint8_t* in_bytes = reinterpret_cast<const uint8_t*>(a.in);
  const uint8_t* zp = zero_page + (lane & 3) * 16;
  unsigned sbase[6], sok = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pp = 16 * (wave + 4 * j) + (lane >> 2);
    const int sy = pp / pw, sx = pp - sy * pw;
    const int iy = oy0 - 1 + sy, ix = sx - 1;
    const bool ok = sy < srows && sx < a.Wo + 2 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
    const int u = (lane & 3) ^ ((sx >> 2) & 3);
    sbase[j] = ok ? (unsigned)((((long)(n_img * a.Hi + iy) * a.Wi + ix) * a.Cin) * 4 + u * 16) : 0u;
    sok |= (ok ? 1u : 0u) << j;
  }
  const uint8_t* wsrc = reinterpret_cast<const uint8_t*>(ab.wdma) + (size_t)(n0 >> 6) * nslots * 4096 + wave * 1024 + lane * 16;
#define SERL_SD_SLAB(J, CG, SB)                                                                                       \
  if (wave + 4 * (J) < kSdPieces) {                                                                                   \
    const uint8_t* src_ = ((sok >> (J)) & 1u) ? in_bytes + (size_t)sbase[J] + ((CG) << 6) : zp;                       \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)src_, (lds_void_t*)(smS + (SB) * kSdSlab + (wave + 4 * (J)) * 1024), 16, 0, 0); \
  }
#define SERL_SD_W1(CG, KY, WB, KX)                                                                                    \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(wsrc + ((size_t)(((KY) * 3 + (KX)) * c16n + (CG)) << 12)),        \
                                     (lds_void_t*)(smW + (WB) * kSdW + (KX) * 4096 + wave * 1024), 16, 0, 0);
#define SERL_SD_W(CG, KY, WB) { SERL_SD_W1(CG, KY, WB, 0) SERL_SD_W1(CG, KY, WB, 1) SERL_SD_W1(CG, KY, WB, 2) }
  f32x16 acc[TM][TN], accx[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; accx[tm][tn][r] = 0.f; }
  const int li = lane & 31, lh = lane >> 5;
  int ahi[TM][3], alo[TM][3];   // LDS byte offsets of this lane's pixel at kernel row 0, per tap column
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = wave * WROWS + tm * 32 + li;
    const int y = r / a.Wo, x = r - y * a.Wo;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int f = ((x + kx) >> 2) & 3, pa = y * pw + x + kx;
      ahi[tm][kx] = pa * 64 + (((2 * lh) ^ f) << 4);
      alo[tm][kx] = pa * 64 + (((2 * lh + 1) ^ f) << 4);
    }
  }
  int bhi[TN], blo[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int row = tn * 32 + li, sw = (row >> 2) & 3;
    bhi[tn] = row * 64 + ((lh ^ sw) << 4);
    blo[tn] = row * 64 + (((2 + lh) ^ sw) << 4);
  }
  // prologue: the whole slab of channel group 0 and the weights of sub-chunk (0, 0)
#pragma unroll
  for (int j = 0; j < 6; ++j) SERL_SD_SLAB(j, 0, 0)
  SERL_SD_W(0, 0, 0)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  int cg = 0, ky = 0;
  for (int c = 0; c < nchunks; ++c) {
    // in flight under this sub-chunk's MFMAs: the weights of the next sub-chunk, two pieces of the next group's slab
    const int ncg = ky == 2 ? cg + 1 : cg, nky = ky == 2 ? 0 : ky + 1;
    const int ncgc = min(ncg, c16n - 1), sn = min(cg + 1, c16n - 1);
    // (all five pieces up front: one piece behind the first MFMA of each tile group -- the ring kernel's placement -- was measured
    //  SLOWER here, 2.382 / 2.375 -> 2.405 / 2.400 ms per step: the late pieces have too few MFMAs left to land behind)
    SERL_SD_W(ncgc, nky, (c + 1) & 1)
    if (ky == 0) { SERL_SD_SLAB(0, sn, (cg + 1) & 1) SERL_SD_SLAB(1, sn, (cg + 1) & 1) }
    else if (ky == 1) { SERL_SD_SLAB(2, sn, (cg + 1) & 1) SERL_SD_SLAB(3, sn, (cg + 1) & 1) }
    else { SERL_SD_SLAB(4, sn, (cg + 1) & 1) SERL_SD_SLAB(5, sn, (cg + 1) & 1) }
    const uint8_t* sa = smS + (cg & 1) * kSdSlab + ky * pw * 64;
    const uint8_t* sb = smW + (c & 1) * kSdW;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      f16x8 fah[TM], fal[TM], fbh[TN], fbl[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        fah[tm] = *reinterpret_cast<const f16x8*>(sa + ahi[tm][kx]);
        fal[tm] = *reinterpret_cast<const f16x8*>(sa + alo[tm][kx]);
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        fbh[tn] = *reinterpret_cast<const f16x8*>(sb + kx * 4096 + bhi[tn]);
        fbl[tn] = *reinterpret_cast<const f16x8*>(sb + kx * 4096 + blo[tn]);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[tm], fbh[tn], accx[tm][tn], 0, 0, 0);
          accx[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[tm], fbl[tn], accx[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[tm], fbh[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    // every DMA issued above has had 36 MFMAs to land; the reads of this sub-chunk are done before anybody refills its buffers
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (++ky == 3) { ky = 0; ++cg; }
  }
#undef SERL_SD_SLAB
#undef SERL_SD_W
  if (EPT) rowtile_epilogue_t(ab, acc, accx, m0, n0, n_img, wave, lane, n_img * a.tiles_n + bn, smemb);
  else rowtile_epilogue(ab, acc, accx, m0, n0, n_img, wave, li, lh);
}

}  // namespace serl
