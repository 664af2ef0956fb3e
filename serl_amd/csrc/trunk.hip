// Frozen ResNet-10 trunk forward for MI355X (gfx950), fp32 in / fp32 accumulate on the f32 MFMA
// (v_mfma_f32_32x32x2_f32).  Reference semantics: serl_launcher/vision/resnet_v1.py:189-286
// (normalise -> conv7x7/2 -> GN -> ReLU -> maxpool3x3/2 SAME -> 4 basic blocks, GroupNorm(4 groups,
// eps 1e-5, fast variance), XLA SAME padding (lo 0, hi 1) for the stride-2 3x3 convs).
//
// Layout: activations NHWC fp32, conv kernels HWIO ( = row-major [K = kh*kw*Cin][Cout] GEMM B operand).
// Every conv is an implicit GEMM  C[M = N*Ho*Wo][Cout] = im2col(A)[M][K] * W[K][Cout]:
//   * 256-thread workgroups, 4 waves, each wave owns a 64x64 output tile = 2x2 MFMA 32x32 tiles;
//   * BK = 32 K-chunks lie inside one (ky,kx) tap, so an A-tile row is one contiguous 128-byte
//     channel segment of the NHWC input (or zeros outside the image) -> 16-byte coalesced loads;
//   * register-staged double-buffered LDS (global loads of chunk c+1 are in flight under the
//     MFMAs of chunk c), LDS row pitch 33 floats: conflict-free ds_read_b32 for the MFMA operands;
//   * the previous layer's GroupNorm+ReLU is applied on load (per-sample scale/shift table), and
//     this layer's GroupNorm statistics are reduced in the epilogue (fp64 atomics per (image,
//     group)), so no activation is re-read just to be normalised.
#include <algorithm>

#include "internal.h"
#include "prof.h"
#include "trunk_common.h"

namespace serl {

// ---------------------------------------------------------------------------------------------
// conv_init: u8 image -> ImageNet normalise -> conv 7x7 stride 2 pad 3, 3 -> 64 (K = 147 -> 148)
// One workgroup = 16x16 output pixels of one image x all 64 output channels.
// ---------------------------------------------------------------------------------------------
struct ConvInitArgs {
  const uint8_t* img;  // [N][H][W][3]
  const float* w;      // [147][64]
  float* out;          // [N][Ho][Wo][64]
  double* stats;       // [N][4][2]
  int N, H, W, Ho, Wo, tiles_y, tiles_x;
};

constexpr int kCiPatch = 37;             // 2*16 + 5 input rows/cols per 16x16 output tile
constexpr int kCiPW = kCiPatch * 3 + 2;  // LDS patch row pitch (floats)
constexpr int kCiK = 148;                // 147 padded to a multiple of the MFMA K (2)
constexpr int kCiBS = 64 + 4;            // weight row pitch in LDS

__global__ __launch_bounds__(256) void conv_init_kernel(ConvInitArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                      // [37][kCiPW]
  float* wl = smem + kCiPatch * kCiPW + 3;  // [148][68]  (+3 keeps 16B alignment: 37*113=4181 -> 4184)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = b % a.tiles_x;
  b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int oy0 = ty * 16, ox0 = tx * 16;
  // weights -> LDS (row 147 zero)
  for (int v = tid; v < kCiK * 16; v += 256) {
    const int row = v >> 4, c4 = v & 15;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < 147) val = *reinterpret_cast<const float4*>(a.w + row * 64 + c4 * 4);
    *reinterpret_cast<float4*>(wl + row * kCiBS + c4 * 4) = val;
  }
  // normalised input patch -> LDS (zero outside the image: the conv pads the NORMALISED tensor)
  const uint8_t* img = a.img + (size_t)n * a.H * a.W * 3;
  const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
  for (int v = tid; v < kCiPatch * kCiPatch * 3; v += 256) {
    const int yy = v / (kCiPatch * 3), rest = v - yy * (kCiPatch * 3);
    const int xx = rest / 3, ch = rest - xx * 3;
    const int iy = iy0 + yy, ix = ix0 + xx;
    float val = 0.f;
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
      const float mean = ch == 0 ? 0.485f : (ch == 1 ? 0.456f : 0.406f);
      const float stdv = ch == 0 ? 0.229f : (ch == 1 ? 0.224f : 0.225f);
      val = ((float)img[((size_t)iy * a.W + ix) * 3 + ch] / 255.0f - mean) / stdv;
    }
    patch[yy * kCiPW + rest] = val;
  }
  __syncthreads();
  // wave `wave` owns output pixels [wave*64, wave*64+64) of the 16x16 tile (row-major), 64 channels
  const int i = lane & 31, h = lane >> 5;
  int abase[2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int p = wave * 64 + tm * 32 + i;
    abase[tm] = (2 * (p >> 4)) * kCiPW + (p & 15) * 6;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
#pragma unroll 2
  for (int ks = 0; ks < kCiK / 2; ++ks) {
    const int k = 2 * ks + h;
    const int kk = k < 147 ? k : 0;  // padded k reads a finite value; its weight row is zero
    const int ky = kk / 21, rest = kk - ky * 21;
    const int koff = ky * kCiPW + rest;
    const float a0 = patch[abase[0] + koff], a1 = patch[abase[1] + koff];
    const float b0 = wl[k * kCiBS + i], b1 = wl[k * kCiBS + 32 + i];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
  // epilogue: raw output + GroupNorm statistics
  float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = wave * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
      const bool ok = oy < a.Ho && ox < a.Wo;
      float* o = a.out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * 64;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const float v = ok ? acc[tm][tn][r] : 0.f;
        if (ok) o[tn * 32 + i] = v;
        s[tn] += v;
        q[tn] += v * v;
      }
    }
  double* st = a.stats + (size_t)n * kGnGroups * 2;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) stats_flush(s[tn], q[tn], st, tn * 32 + i, 16, true);
}

// fallback GroupNorm statistics pass (used when the conv epilogue cannot attribute its rows to
// images, i.e. Ho*Wo is neither a multiple of 64 nor 16/32): one workgroup per (image, group).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* x, double* stats, int P, int Cc) {
  const int n = blockIdx.x / kGnGroups, g = blockIdx.x % kGnGroups;
  const int gs = Cc / kGnGroups;
  const float* xb = x + (size_t)n * P * Cc + g * gs;
  double s = 0.0, q = 0.0;
  for (int e = threadIdx.x; e < P * gs; e += 256) {
    const int p = e / gs, c = e - p * gs;
    const float v = xb[(size_t)p * Cc + c];
    s += v;
    q += (double)v * v;
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[((size_t)n * kGnGroups + g) * 2] = red[0][0];
    stats[((size_t)n * kGnGroups + g) * 2 + 1] = red[1][0];
  }
}

// ---------------------------------------------------------------------------------------------
// GN-apply + ReLU + max_pool 3x3 stride 2 SAME (pad lo 0 / hi 1 with -inf)   (resnet_v1.py:257-259)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_relu_maxpool_kernel(const float* x, GnRef gn, float* out, int N,
                                                             int Hi, int Wi, int Ho, int Wo, int Cc) {
  const int c4n = Cc / 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * Ho * Wo * c4n) return;
  const int c4 = (int)(e % c4n);
  long t = e / c4n;
  const int ox = (int)(t % Wo);
  t /= Wo;
  const int oy = (int)(t % Ho);
  const int n = (int)(t / Ho);
  float4 s, h;
  gn_coef4(gn, n, c4 * 4, s, h);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 + dy;
    if (iy >= Hi) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 + dx;
      if (ix >= Wi) continue;
      const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * Hi + iy) * Wi + ix) * Cc + c4 * 4);
      m.x = fmaxf(m.x, fmaxf(v.x * s.x + h.x, 0.f));
      m.y = fmaxf(m.y, fmaxf(v.y * s.y + h.y, 0.f));
      m.z = fmaxf(m.z, fmaxf(v.z * s.z + h.z, 0.f));
      m.w = fmaxf(m.w, fmaxf(v.w * s.w + h.w, 0.f));
    }
  }
  *reinterpret_cast<float4*>(out + (((size_t)n * Ho + oy) * Wo + ox) * Cc + c4 * 4) = m;
}

// block output: out = relu( GN(raw_b) + residual ), residual = res (already activated) or GN(res_raw)
__global__ __launch_bounds__(256) void block_out_kernel(const float* raw, GnRef gn, const float* res, GnRef rgn,
                                                       float* out, int N, int P, int Cc) {
  const int c4n = Cc / 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * P * c4n) return;
  const int c4 = (int)(e % c4n);
  const int n = (int)(e / ((long)P * c4n));
  const float4 v = reinterpret_cast<const float4*>(raw)[e];
  float4 s, h;
  gn_coef4(gn, n, c4 * 4, s, h);
  float4 r = reinterpret_cast<const float4*>(res)[e];
  if (rgn.stats) {
    float4 s2, h2;
    gn_coef4(rgn, n, c4 * 4, s2, h2);
    r.x = r.x * s2.x + h2.x; r.y = r.y * s2.y + h2.y; r.z = r.z * s2.z + h2.z; r.w = r.w * s2.w + h2.w;
  }
  float4 o;
  o.x = fmaxf(r.x + (v.x * s.x + h.x), 0.f);
  o.y = fmaxf(r.y + (v.y * s.y + h.y), 0.f);
  o.z = fmaxf(r.z + (v.z * s.z + h.z), 0.f);
  o.w = fmaxf(r.w + (v.w * s.w + h.w), 0.f);
  reinterpret_cast<float4*>(out)[e] = o;
}

// ---------------------------------------------------------------------------------------------
// generic implicit-GEMM conv (3x3 / 1x1, Cin % 32 == 0, Cout % (64*WN) == 0)
// ---------------------------------------------------------------------------------------------

// PMODE: how output rows map to images for the GN statistics
//   0: P % (32*TM) == 0 (a wave's rows lie in one image)   1: P == 32   2: P == 16   3: no stats
// TM/TN: 32x32 MFMA tiles per wave along M/N.  Workgroup tile = (32*TM*WM) x (32*TN*WN).
template <int WM, int WN, int TM, int TN, int PMODE, bool ONE_IMG>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int WROWS = 32 * TM;
  constexpr int WCOLS = 32 * TN;
  constexpr int BM = WROWS * WM, BN = WCOLS * WN, AS = 33, BS = BN + 4;
  constexpr int AI = BM / 32;  // float4 A loads per thread per chunk
  constexpr int BI = BN / 32;  // float4 B loads per thread per chunk
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                // [2][BM][AS]
  float* Bs = smem + 2 * BM * AS;  // [2][32][BS]
  float* gnm = Bs + 2 * 32 * BS;   // [(BM+1)*4] mean  of (image in tile, group) of the producing layer
  float* gnr = gnm + (BM + 1) * 4; // [(BM+1)*4] rstd
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bn = id % a.tiles_n, bm = id / a.tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;
  const int n_first = m0 / a.P;
  if (a.in_gn.stats) {  // GroupNorm statistics of the producer -> (mean, rstd) table for this tile's images
    const int n_last = min(a.N - 1, (m0 + BM - 1) / a.P);
    for (int t = tid; t < (n_last - n_first + 1) * kGnGroups; t += 256) {
      const double* st = a.in_gn.stats + ((size_t)n_first * kGnGroups + t) * 2;
      const double mean = st[0] * a.in_gn.inv_count, m2 = st[1] * a.in_gn.inv_count;
      gnm[t] = (float)mean;
      gnr[t] = rsqrtf(fmaxf((float)(m2 - mean * mean), 0.f) + 1e-5f);
    }
    __syncthreads();
  }

  // ---- per-thread im2col row bookkeeping (rows are fixed across K chunks)
  const int kq = tid & 7;
  int rpix[AI], riy[AI], rix[AI], rn[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int m = m0 + (tid >> 3) + 32 * i;
    if (m < a.M) {
      const int n = m / a.P, rem = m - n * a.P;
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      rn[i] = n;
      rpix[i] = n * a.Hi * a.Wi;
      riy[i] = oy * a.stride - a.pad;
      rix[i] = ox * a.stride - a.padw;
    } else {
      rn[i] = 0; rpix[i] = 0; riy[i] = -(1 << 20); rix[i] = -(1 << 20);
    }
  }
  // ONE_IMG: one image per workgroup tile (P % BM == 0) -> one scale/shift vector per chunk serves all
  // rows of this thread.  (Compile-time: a runtime select over the arrays would put them in scratch.)
  const int cpt = a.Cin >> 5;  // 32-channel chunks per tap
  const int nchunks = a.KH * a.KW * cpt;
  constexpr int BQ = BN / 4;            // float4 per weight-tile row
  constexpr int BROWSTEP = 256 / BQ;    // rows covered per pass of the 256 threads
  const int brow = tid / BQ, bcol = tid % BQ;

  constexpr int NS = ONE_IMG ? 1 : AI;
  float4 ra[AI], rb[BI], rs[NS], rh[NS];
  unsigned okmask = 0;
  // Loads are unconditional (coordinates clamped into the image): nothing consumes a loaded value
  // before the MFMA phase of the current chunk is over, so all of them stay in flight together.
  // (Macros, not lambdas: by-reference captures kept the staging arrays in scratch memory.)
#define SERL_LOAD_CHUNK(CIDX)                                                                                  \
  {                                                                                                            \
    const int c_ = (CIDX);                                                                                     \
    const int tap = c_ / cpt, ci0 = (c_ - tap * cpt) << 5;                                                     \
    const int ky = tap / a.KW, kx = tap - ky * a.KW;                                                           \
    okmask = 0;                                                                                                \
    _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                           \
      const int iy = riy[i] + ky, ix = rix[i] + kx;                                                            \
      const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;                          \
      okmask |= (ok ? 1u : 0u) << i;                                                                           \
      const int cy = min(max(iy, 0), a.Hi - 1), cx = min(max(ix, 0), a.Wi - 1);                                \
      ra[i] = *reinterpret_cast<const float4*>(a.in + (size_t)(rpix[i] + cy * a.Wi + cx) * a.Cin + ci0 + 4 * kq); \
    }                                                                                                          \
    if (a.in_gn.stats) {                                                                                       \
      const int c4_ = ci0 + 4 * kq, grp_ = c4_ / a.in_gn.gsize;                                                \
      const float4 ga_ = *reinterpret_cast<const float4*>(a.in_gn.gamma + c4_);                                \
      const float4 be_ = *reinterpret_cast<const float4*>(a.in_gn.beta + c4_);                                 \
      _Pragma("unroll") for (int i = 0; i < NS; ++i) {                                                         \
        const int t_ = max(rn[i] - n_first, 0) * kGnGroups + grp_;                                             \
        const float mean_ = gnm[t_], rstd_ = gnr[t_];                                                          \
        rs[i] = make_float4(ga_.x * rstd_, ga_.y * rstd_, ga_.z * rstd_, ga_.w * rstd_);                       \
        rh[i] = make_float4(be_.x - mean_ * rs[i].x, be_.y - mean_ * rs[i].y, be_.z - mean_ * rs[i].z,         \
                            be_.w - mean_ * rs[i].w);                                                          \
      }                                                                                                        \
    }                                                                                                          \
    const float* wp = a.w + (size_t)(c_ << 5) * a.Cout + n0 + 4 * bcol;                                        \
    _Pragma("unroll") for (int i = 0; i < BI; ++i)                                                             \
        rb[i] = *reinterpret_cast<const float4*>(wp + (size_t)(brow + BROWSTEP * i) * a.Cout);                 \
  }
  // GroupNorm+ReLU of the producing layer and the zero padding are applied here, on the way to LDS
#define SERL_STORE_CHUNK(BUF)                                                                                  \
  {                                                                                                            \
    float* Ab_ = As + (BUF) * BM * AS;                                                                         \
    float* Bb_ = Bs + (BUF) * 32 * BS;                                                                         \
    _Pragma("unroll") for (int i = 0; i < AI; ++i) {                                                           \
      float4 v = ra[i];                                                                                        \
      if (a.in_gn.stats) {                                                                                     \
        const float4 s_ = rs[ONE_IMG ? 0 : i];                                                                 \
        const float4 h_ = rh[ONE_IMG ? 0 : i];                                                                 \
        v.x = fmaxf(v.x * s_.x + h_.x, 0.f);                                                                   \
        v.y = fmaxf(v.y * s_.y + h_.y, 0.f);                                                                   \
        v.z = fmaxf(v.z * s_.z + h_.z, 0.f);                                                                   \
        v.w = fmaxf(v.w * s_.w + h_.w, 0.f);                                                                   \
      }                                                                                                        \
      if (!((okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);                                          \
      float* p_ = Ab_ + ((tid >> 3) + 32 * i) * AS + 4 * kq;                                                   \
      p_[0] = v.x; p_[1] = v.y; p_[2] = v.z; p_[3] = v.w;                                                      \
    }                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < BI; ++i)                                                             \
        *reinterpret_cast<float4*>(Bb_ + (brow + BROWSTEP * i) * BS + 4 * bcol) = rb[i];                       \
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  SERL_LOAD_CHUNK(0);
  SERL_STORE_CHUNK(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    // unconditional prefetch/store (the last iteration re-loads its own chunk into the idle buffer):
    // keeps the staging registers in SSA form -- a conditional here makes hipcc spill them to scratch
    // and wait for every load right after issuing it.
    SERL_LOAD_CHUNK(min(c + 1, nchunks - 1));
    const float* Ab = As + buf * BM * AS + (wm * WROWS + li) * AS + lh;
    const float* Bb = Bs + buf * 32 * BS + lh * BS + wn * WCOLS + li;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      float bv[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bv[tn] = Bb[2 * ks * BS + 32 * tn];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const float av = Ab[tm * 32 * AS + 2 * ks];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[tn], acc[tm][tn], 0, 0, 0);
      }
    }
    SERL_STORE_CHUNK(buf ^ 1);
    __syncthreads();
  }
#undef SERL_LOAD_CHUNK
#undef SERL_STORE_CHUNK

  // ---- epilogue: raw conv output (NHWC == row-major [M][Cout])
  const int wrow0 = m0 + wm * WROWS;
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
  // ---- epilogue: GroupNorm statistics (rows beyond M are exact zeros and contribute nothing)
  if (PMODE != 3) {
    const int gsize = a.Cout / kGnGroups;
    constexpr int ROWS = PMODE == 0 ? WROWS : (PMODE == 1 ? 32 : 16);  // rows per image slot
    constexpr int NSLOT = WROWS / ROWS;
#pragma unroll
    for (int slot = 0; slot < NSLOT; ++slot) {
      const int mrow = wrow0 + slot * ROWS;
      const bool valid = mrow < a.M;
      const int n = valid ? mrow / a.P : 0;
      double* st = a.stats + (size_t)n * kGnGroups * 2;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = tm * 32 + 8 * (r >> 2);  // (+ (r&3) + 4*lh < 8): 8-row granules
            if (row / ROWS == slot) {
              const float v = acc[tm][tn][r];
              s += v;
              q += v * v;
            }
          }
        stats_flush(s, q, st, n0 + wn * WCOLS + tn * 32 + li, gsize, valid);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline int cd(int a, int b) { return (a + b - 1) / b; }

TrunkDims trunk_dims(int H, int W) {
  TrunkDims d{};
  d.H = H; d.W = W;
  d.h[0] = cd(H, 2); d.w[0] = cd(W, 2);
  d.h[1] = cd(d.h[0], 2); d.w[1] = cd(d.w[0], 2);
  for (int i = 0; i < kTrunkStages; ++i) {
    d.h[2 + i] = cd(d.h[1 + i], kStageStride[i]);
    d.w[2 + i] = cd(d.w[1 + i], kStageStride[i]);
  }
  return d;
}

constexpr int kGnLayers = 1 + 3 * kTrunkStages;
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t trunk_layout(TrunkWorkspace* ws, uint8_t* base, int N, int H, int W) {
  const TrunkDims d = trunk_dims(H, W);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    uint8_t* p = base ? base + off : nullptr;
    off += al256(bytes);
    return p;
  };
  float* raw_init = (float*)take((size_t)N * d.h[0] * d.w[0] * 64 * 4);
  float* pool = (float*)take((size_t)N * d.h[1] * d.w[1] * 64 * 4);
  TrunkWorkspace::B blk[kTrunkStages];
  for (int i = 0; i < kTrunkStages; ++i) {
    const size_t e = (size_t)N * d.h[2 + i] * d.w[2 + i] * kStageFilters[i] * 4;
    blk[i].raw0 = (float*)take(e);
    blk[i].raw1 = (float*)take(e);
    blk[i].rawp = (float*)take(e);
    blk[i].out = (float*)take(e);
    blk[i].norm0 = (float*)take(e);
  }
  float* kslab = (float*)take((size_t)kKsplitTiles * 64 * 64 * 4);
  int* kctr = (int*)take((size_t)kKsplitTiles * sizeof(int));
  const size_t stats_bytes = al256((size_t)kGnLayers * N * kGnGroups * 2 * sizeof(double));
  const size_t sync_bytes = (size_t)kGnLayers * ((size_t)N * kSyncPerImage + kSyncTickets) * sizeof(int);
  double* stats = (double*)take(stats_bytes + sync_bytes);
  if (ws) {
    ws->sync = base ? (int*)(base + ((uint8_t*)stats - base) + stats_bytes) : nullptr;
    ws->stats_sync_bytes = stats_bytes + sync_bytes;
    ws->max_images = N; ws->d = d; ws->raw_init = raw_init; ws->pool = pool;
    for (int i = 0; i < kTrunkStages; ++i) ws->blk[i] = blk[i];
    ws->stats = stats; ws->base = base; ws->bytes = off;
    ws->kslab = kslab; ws->kctr = kctr;
  }
  return off;
}

size_t trunk_workspace_bytes(int max_images, int H, int W) {
  return trunk_layout(nullptr, nullptr, max_images, H, W);
}

int trunk_workspace_bind(TrunkWorkspace& ws, void* mem, int max_images, int H, int W) {
  SERL_REQUIRE(mem != nullptr, "trunk workspace memory is NULL");
  SERL_REQUIRE(H >= 32 && W >= 32, "trunk needs images of at least 32x32 (got %dx%d)", H, W);
  trunk_layout(&ws, (uint8_t*)mem, max_images, H, W);
  return SERL_OK;
}

static GnRef gn_ref(const double* stats, const float* gamma, const float* beta, int P, int Cc) {
  GnRef g{};
  g.stats = stats; g.gamma = gamma; g.beta = beta;
  g.inv_count = 1.0 / ((double)P * (Cc / kGnGroups));
  g.gsize = Cc / kGnGroups;
  return g;
}

static int launch_conv(const char* tag, const float* in, const float* w, float* out, double* stats, GnRef in_gn,
                       int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksz, int stride,
                       hipStream_t stream) {
  SERL_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "conv channels unsupported (Cin %d, Cout %d)", Cin, Cout);
  ConvArgs a{};
  a.in = in; a.w = w; a.out = out; a.stats = stats; a.in_gn = in_gn;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.KH = a.KW = ksz; a.stride = stride;
  // XLA SAME padding: total = max((ceil(n/s)-1)*s + k - n, 0), lo = total/2
  a.pad = std::max((Ho - 1) * stride + ksz - Hi, 0) / 2;
  a.padw = std::max((Wo - 1) * stride + ksz - Wi, 0) / 2;
  a.M = N * Ho * Wo; a.P = Ho * Wo;
  // tile choice: 128x128 (2x2 waves of 64x64) for wide layers, 128x64 (4x1 waves of 32x64) for Cout == 64,
  // 64x64 (2x2 waves of 32x32) when the big tile would leave most of the 256 CUs idle (small batches,
  // i.e. one rank's share of a data-parallel job).
  int cfg = Cout >= 128 ? 0 : 1;
  if (cfg == 0 && (long)cd(a.M, 128) * (Cout / 128) < 512) cfg = 2;
  const int BM = cfg == 2 ? 64 : 128, BN = cfg == 0 ? 128 : 64;
  const int wrows = cfg == 0 ? 64 : 32;
  a.tiles_m = cd(a.M, BM); a.tiles_n = Cout / BN;
  const size_t lds = (size_t)(2 * BM * 33 + 2 * 32 * (BN + 4) + 2 * (BM + 1) * 4) * 4;
  int pmode = (a.P % wrows == 0) ? 0 : (a.P == 32 ? 1 : (a.P == 16 ? 2 : 3));
  if (cfg != 0 && pmode == 1) pmode = 3;
  const bool one_img = (a.P % BM) == 0;
  dim3 grid(a.tiles_m * a.tiles_n), block(256);
  {
    ProfScope prof(tag, stream);
#define SERL_LAUNCH_CONV2(WM, WN, TM, TN, PM)                                                                      \
  do {                                                                                                             \
    if (one_img) hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, PM, true>), grid, block, lds, stream, a);   \
    else hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, PM, false>), grid, block, lds, stream, a);          \
  } while (0)
#define SERL_LAUNCH_CONV(WM, WN, TM, TN)                       \
  do {                                                         \
    if (pmode == 0) SERL_LAUNCH_CONV2(WM, WN, TM, TN, 0);      \
    else if (pmode == 1) SERL_LAUNCH_CONV2(WM, WN, TM, TN, 1); \
    else if (pmode == 2) SERL_LAUNCH_CONV2(WM, WN, TM, TN, 2); \
    else SERL_LAUNCH_CONV2(WM, WN, TM, TN, 3);                 \
  } while (0)
    if (cfg == 0) SERL_LAUNCH_CONV(2, 2, 2, 2);
    else if (cfg == 1) SERL_LAUNCH_CONV(4, 1, 1, 2);
    else SERL_LAUNCH_CONV(2, 2, 1, 1);
#undef SERL_LAUNCH_CONV
#undef SERL_LAUNCH_CONV2
  }
  SERL_HIP(hipGetLastError());
  if (pmode == 3) {  // statistics in a separate pass
    hipLaunchKernelGGL(gn_stats_kernel, dim3(N * kGnGroups), dim3(256), 0, stream, out, stats, a.P, Cout);
    SERL_HIP(hipGetLastError());
  }
  return SERL_OK;
}

static void conv_dims(int i, int which, int& K, int& Cout) {
  const int cin = i == 0 ? 64 : kStageFilters[i - 1], f = kStageFilters[i];
  Cout = f;
  K = which == 0 ? 9 * cin : (which == 1 ? 9 * f : cin);
}

// layers the LDS-DMA kernel may take (at least 128 output channels, K a multiple of 32)
static bool dma_copy(int i, int which) { return kStageFilters[i] >= 128 || (i == 0 && which <= 1); }   // (+ block 0: the row-slab kernels' LDS-DMA weight / operand staging)

size_t trunk_packed_bytes() {
  size_t off = 0;
  for (int i = 0; i < kTrunkStages; ++i)
    for (int k = 0; k < 3; ++k) {
      if (k == 2 && i == 0) continue;
      int K, Cout;
      conv_dims(i, k, K, Cout);
      off += 2 * al256((size_t)K * Cout * 2) + al256((size_t)Cout * 4);
      if (dma_copy(i, k)) off += al256((size_t)2 * K * Cout * 2);
    }
  off += 2 * al256((size_t)64 * 224 * 2) + al256(64 * 4);   // conv_init planes (u8 variant: [64][224]; the 3-product variant uses [64][176] of it)
  off += 256;                                // zero page (the agent arena is zero-initialised and nothing writes here)
  return off;
}

int trunk_packed_bind(TrunkPacked& p, void* mem) {
  SERL_REQUIRE(mem != nullptr, "packed-weight memory is NULL");
  uint8_t* b = (uint8_t*)mem;
  size_t off = 0;
  for (int i = 0; i < kTrunkStages; ++i)
    for (int k = 0; k < 3; ++k) {
      if (k == 2 && i == 0) continue;
      int K, Cout;
      conv_dims(i, k, K, Cout);
      p.blk[i][k].hi = (uint16_t*)(b + off); off += al256((size_t)K * Cout * 2);
      p.blk[i][k].lo = (uint16_t*)(b + off); off += al256((size_t)K * Cout * 2);
      p.blk[i][k].inv = (float*)(b + off); off += al256((size_t)Cout * 4);
      p.blk[i][k].dma = nullptr;
      if (dma_copy(i, k)) { p.blk[i][k].dma = (uint16_t*)(b + off); off += al256((size_t)2 * K * Cout * 2); }
    }
  p.init.hi = (uint16_t*)(b + off); off += al256((size_t)64 * 224 * 2);
  p.init.lo = (uint16_t*)(b + off); off += al256((size_t)64 * 224 * 2);
  p.init.inv = (float*)(b + off); off += al256(64 * 4);
  p.zero = b + off; off += 256;
  p.dirty = true;
  return SERL_OK;
}

static int trunk_pack(const TrunkWeights& w, TrunkPacked& p, hipStream_t stream) {
  for (int i = 0; i < kTrunkStages; ++i)
    for (int k = 0; k < 3; ++k) {
      const float* src = k == 0 ? w.blk[i].conv0 : (k == 1 ? w.blk[i].conv1 : w.blk[i].proj);
      if (!src) continue;
      int K, Cout;
      conv_dims(i, k, K, Cout);
      int rc = pack_conv_weights_f16x3(src, p.blk[i][k].hi, p.blk[i][k].lo, p.blk[i][k].inv, K, Cout, stream);
      if (rc) return rc;
      if (p.blk[i][k].dma && (rc = pack_dma_order_f16x3(p.blk[i][k].hi, p.blk[i][k].lo, p.blk[i][k].dma, K, Cout, stream))) return rc;
    }
  {
    int rc = pack_conv_init_f16x3(w.conv_init, p.init.hi, p.init.lo, p.init.inv, stream);
    if (rc) return rc;
  }
  p.dirty = false;
  return SERL_OK;
}

int trunk_forward(const TrunkWeights& w, TrunkWorkspace& ws, const uint8_t* frames, int n, float* feats_out,
                  hipStream_t stream, TrunkPacked* packed, int stage_begin, int stage_end) {
  if (packed && packed->dirty) {
    int prc = trunk_pack(w, *packed, stream);
    if (prc) return prc;
  }
  if (packed) {
    SERL_REQUIRE(n > 0 && n <= ws.max_images, "trunk_forward: %d images exceeds workspace (%d)", n, ws.max_images);
    return trunk_forward_f16x3(w, ws, *packed, frames, n, feats_out, stream, stage_begin, stage_end);
  }
  SERL_REQUIRE(stage_begin < 0 && stage_end == kTrunkStages - 1, "the exact-fp32 trunk runs whole passes only");
  SERL_REQUIRE(n > 0 && n <= ws.max_images, "trunk_forward: %d images exceeds workspace (%d)", n, ws.max_images);
  const TrunkDims& d = ws.d;
  const int N = n;
  auto stats_of = [&](int layer) { return ws.stats + (size_t)layer * ws.max_images * kGnGroups * 2; };
  SERL_HIP(hipMemsetAsync(ws.stats, 0, (size_t)kGnLayers * ws.max_images * kGnGroups * 2 * sizeof(double), stream));
  int rc;
  {  // conv_init, exact fp32 MFMA
    ConvInitArgs a{};
    a.img = frames; a.w = w.conv_init; a.out = ws.raw_init; a.stats = stats_of(0);
    a.N = N; a.H = d.H; a.W = d.W; a.Ho = d.h[0]; a.Wo = d.w[0];
    a.tiles_y = cd(a.Ho, 16); a.tiles_x = cd(a.Wo, 16);
    const size_t lds = (size_t)(kCiPatch * kCiPW + 3 + kCiK * kCiBS) * 4;
    ProfScope prof("conv_init", stream);
    hipLaunchKernelGGL(conv_init_kernel, dim3(N * a.tiles_y * a.tiles_x), dim3(256), lds, stream, a);
    SERL_HIP(hipGetLastError());
  }
  {
    const long tot = (long)N * d.h[1] * d.w[1] * 16;
    ProfScope prof("gn_relu_maxpool", stream);
    hipLaunchKernelGGL(gn_relu_maxpool_kernel, dim3(cd(tot, 256)), dim3(256), 0, stream, ws.raw_init,
                       gn_ref(stats_of(0), w.gn_init_s, w.gn_init_b, d.h[0] * d.w[0], 64), ws.pool, N, d.h[0],
                       d.w[0], d.h[1], d.w[1], 64);
    SERL_HIP(hipGetLastError());
  }
  const float* x = ws.pool;
  int cin = 64;
  const GnRef none{};
  for (int i = 0; i < kTrunkStages; ++i) {
    const int f = kStageFilters[i], s = kStageStride[i];
    const int Hi = d.h[1 + i], Wi = d.w[1 + i], Ho = d.h[2 + i], Wo = d.w[2 + i], P = Ho * Wo;
    const int l0 = 1 + 3 * i, l1 = 2 + 3 * i, lp = 3 + 3 * i;
    const TrunkWeights::Block& bw = w.blk[i];
    const bool has_proj = bw.proj != nullptr;
    static const char* kTags[kTrunkStages][3] = {{"conv_igemm/b0_conv0", "conv_igemm/b0_conv1", "conv_igemm/b0_proj"},
                                                  {"conv_igemm/b1_conv0", "conv_igemm/b1_conv1", "conv_igemm/b1_proj"},
                                                  {"conv_igemm/b2_conv0", "conv_igemm/b2_conv1", "conv_igemm/b2_proj"},
                                                  {"conv_igemm/b3_conv0", "conv_igemm/b3_conv1", "conv_igemm/b3_proj"}};
    auto conv = [&](int which, const float* in, const float* wf, float* out, double* st, GnRef g, int hi, int wi,
                    int ci, int ksz, int strd) -> int {
      return launch_conv(kTags[i][which], in, wf, out, st, g, N, hi, wi, ci, Ho, Wo, f, ksz, strd, stream);
    };
    if ((rc = conv(0, x, bw.conv0, ws.blk[i].raw0, stats_of(l0), none, Hi, Wi, cin, 3, s))) return rc;
    if (has_proj)
      if ((rc = conv(2, x, bw.proj, ws.blk[i].rawp, stats_of(lp), none, Hi, Wi, cin, 1, s))) return rc;
    if ((rc = conv(1, ws.blk[i].raw0, bw.conv1, ws.blk[i].raw1, stats_of(l1),
                   gn_ref(stats_of(l0), bw.gn0_s, bw.gn0_b, P, f), Ho, Wo, f, 3, 1))) return rc;
    float* out = (i == kTrunkStages - 1) ? feats_out : ws.blk[i].out;
    const long tot = (long)N * P * (f / 4);
    ProfScope prof("block_out", stream);
    hipLaunchKernelGGL(block_out_kernel, dim3(cd(tot, 256)), dim3(256), 0, stream, ws.blk[i].raw1,
                       gn_ref(stats_of(l1), bw.gn1_s, bw.gn1_b, P, f), has_proj ? ws.blk[i].rawp : x,
                       has_proj ? gn_ref(stats_of(lp), bw.gnp_s, bw.gnp_b, P, f) : none, out, N, P, f);
    SERL_HIP(hipGetLastError());
    x = out;
    cin = f;
  }
  return SERL_OK;
}

}  // namespace serl
