// Trainable SmallEncoder: forward, and backward to every conv kernel / bias, on MI355X.
// Reference: serl_launcher/vision/small_encoders.py:9-55 (features (32,64,128,256), 3x3 kernels, stride 2, padding
// VALID, pool_method "avg" -- agents/continuous/drq.py:137-153).
//
// A conv layer is a GEMM of the update chain's kernel (heads.hip gemm_bf16x3: fp32 operands split exactly into three bf16
// pieces, fp32 accumulate) over the layer's im2col matrix, which carries a trailing column of ones; a layer's parameters sit
// in the arena as [9*cin + 1][cout] (HWIO kernel immediately followed by the bias), so the bias is part of the forward GEMM
// and its gradient part of the weight-gradient GEMM, which writes straight into the gradient arena.  Layers 1..3 are IMPLICIT
// GEMMs: the im2col matrix is never written -- the GEMM's operand loader gathers a patch row from the NHWC activations (one
// kernel row = 3*cin contiguous floats; GemmDesc::gtab holds every row's patch offset), forward and weight gradient alike, and
// ReLU is the forward GEMM's epilogue.  (Round 2 wrote explicit im2col matrices: 1.2 GB per pass, 46 % of the step.)  Layer 0
// (u8 frames, K = 27) runs on the frames directly with fp32 FMAs, forward and weight gradient (round 4: small_conv0_*_kernel);
// the input gradient of layers 1..3 is dcol = dy x W^T followed by a col2im gather.
#include <algorithm>
#include <cstdlib>

#include "heads.h"
#include "prof.h"
#include "small_encoder.h"

namespace serl {

SmallDims small_dims(int H, int W) {
  SmallDims d{};
  d.H = H; d.W = W;
  d.h[0] = H; d.w[0] = W;
  for (int l = 0; l < kSmallLayers; ++l) {   // VALID, kernel 3, stride 2
    d.h[l + 1] = (d.h[l] - 3) / 2 + 1;
    d.w[l + 1] = (d.w[l] - 3) / 2 + 1;
  }
  return d;
}
long small_conv_offset(int layer) {
  long off = 0;
  for (int l = 0; l < layer; ++l) off += (9L * kSmallFeat[l] + 1) * kSmallFeat[l + 1];
  return off;
}
long small_conv_params() { return small_conv_offset(kSmallLayers); }

constexpr int kConv0Chunks = 1024, kSmallMaxCams = 4;   // layer 0's weight gradient: row chunks per camera / cameras
static int ldk(int l) { return (9 * kSmallFeat[l] + 1 + 3) & ~3; }   // row pitch of col_l: K + 1 rounded up to 4 floats
static long rows_of(const SmallDims& d, int l, long n_img) { return n_img * d.h[l + 1] * d.w[l + 1]; }
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t carve(SmallWorkspace& ws, uint8_t* base, int max_images, int H, int W) {
  ws.d = small_dims(H, W);
  ws.max_images = max_images;
  size_t off = 0;
  auto take = [&](size_t floats) { float* p = base ? (float*)(base + off) : nullptr; off += al256(floats * 4); return p; };
  long max_act = 0, max_col = 0;
  for (int l = 0; l < kSmallLayers; ++l) {
    const long r = rows_of(ws.d, l, max_images);
    if (l > 0) ws.tab[l] = reinterpret_cast<int*>(take((size_t)r));
    ws.act[l] = take((size_t)r * kSmallFeat[l + 1]);
    max_act = std::max(max_act, r * kSmallFeat[l + 1]);
    if (l > 0) max_col = std::max(max_col, r * ldk(l));
  }
  ws.tab_ready = false;
  ws.dact = take(max_act);
  ws.dact2 = take(max_act);
  ws.dcol = take(max_col);
  ws.slabs_cap = std::max(64L * (9 * 128 + 1) * 256, (long)kConv0Chunks * kSmallMaxCams * 28 * 32);   // up to 64 K-slices of the largest [K+1][cout] gradient / layer 0's row chunks
  ws.slabs = take(ws.slabs_cap);
  ws.bytes = off;
  return off;
}
size_t small_workspace_bytes(int max_images, int H, int W) {
  SmallWorkspace t;
  return carve(t, nullptr, max_images, H, W);
}
int small_workspace_bind(SmallWorkspace& ws, void* mem, int max_images, int H, int W) {
  carve(ws, (uint8_t*)mem, max_images, H, W);
  return SERL_OK;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// col[m][(ky*3+kx)*cin + ci] = x[n][2*oy+ky][2*ox+kx][ci] (u8 input: /255, small_encoders.py:23), col[m][9*cin] = 1,
// padding columns up to the pitch = 0.  One thread per (row m, tap).
template <bool U8>
__global__ __launch_bounds__(256) void small_im2col_kernel(const void* xin, float* col, long rows, int hi, int wi, int ho,
                                                          int wo, int cin, int pitch, int n_per_cam, long cam_stride_img) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * 10) return;
  const long m = e / 10;
  const int tap = (int)(e - m * 10);
  float* dst = col + m * pitch;
  if (tap == 9) {   // ones column (the bias) + zero padding of the pitch
    dst[9 * cin] = 1.0f;
    for (int k = 9 * cin + 1; k < pitch; ++k) dst[k] = 0.0f;
    return;
  }
  const long n = m / ((long)ho * wo);
  const int rem = (int)(m - n * (long)ho * wo), oy = rem / wo, ox = rem - oy * wo;
  const int ky = tap / 3, kx = tap - 3 * ky;
  // (the u8 frames of a camera may be a slice of a larger batch: camera blocks are cam_stride_img images apart)
  const long nn = U8 ? (n / n_per_cam) * cam_stride_img + (n % n_per_cam) : n;
  const long src = ((nn * hi + 2 * oy + ky) * wi + 2 * ox + kx) * cin;
  if (U8) {
    const uint8_t* p = static_cast<const uint8_t*>(xin) + src;
    for (int c = 0; c < cin; ++c) dst[tap * cin + c] = (float)p[c] / 255.0f;
  } else {
    const float* p = static_cast<const float*>(xin) + src;
    for (int c = 0; c < cin; c += 4) *reinterpret_cast<float4*>(dst + tap * cin + c) = *reinterpret_cast<const float4*>(p + c);
  }
}

// ---------------------------------------------------------------------------------------------
// Layer 0 (3 -> 32 channels, K = 27) directly on the u8 frames, fp32 FMAs: no im2col matrix (round 3 wrote and re-read 450 MB
// of it per forward pass: an im2col launch of 120 us + a K = 28 GEMM whose bf16 split costs more than its MFMAs).
// Forward: a thread = (output pixel, 8 output channels); the 28 x 32 parameter block ([27 taps][32] kernel + bias row) sits in
// LDS, a thread reads its 3 x 9 bytes with three 8-byte + three 1-byte loads.  act = relu(bias + sum_t px_t / 255 * w_t).
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) U64Unaligned { unsigned long long v; };

__global__ __launch_bounds__(256) void small_conv0_fwd_kernel(const uint8_t* frames, const float* P, long cam_stride, float* act,
                                                             long rows_cam, int n_per_cam, long frame_cam_stride, int hi, int wi,
                                                             int ho, int wo) {
  __shared__ float Ws[28 * 32];
  __shared__ float lut[256];   // byte -> byte / 255.0f (the division's exact result, once per workgroup instead of 27x per thread)
  const int cam = blockIdx.y, tid = threadIdx.x;
  const float* Wg = P + (long)cam * cam_stride;
  for (int i = tid; i < 28 * 32; i += 256) Ws[i] = Wg[i];
  lut[tid] = (float)tid / 255.0f;
  __syncthreads();
  const long e = (long)blockIdx.x * 256 + tid;
  const long m = e >> 2;
  const int cg = (int)(e & 3) * 8;
  if (m >= rows_cam) return;
  const long n = m / ((long)ho * wo);
  const int rem = (int)(m - n * (long)ho * wo), oy = rem / wo, ox = rem - oy * wo;
  const uint8_t* px = frames + ((((long)cam * frame_cam_stride + n) * hi + 2 * oy) * wi + 2 * ox) * 3;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = Ws[27 * 32 + cg + j];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const uint8_t* q = px + (long)ky * wi * 3;
    const unsigned long long lo = reinterpret_cast<const U64Unaligned*>(q)->v;
    const unsigned b8 = q[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float x = lut[t < 8 ? (unsigned)((lo >> (8 * t)) & 0xffu) : b8];   // x / 255 (small_encoders.py:23)
      const float* w = Ws + (ky * 9 + t) * 32 + cg;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += x * w[j];
    }
  }
  float* o = act + ((long)cam * rows_cam + m) * 32 + cg;
  *reinterpret_cast<float4*>(o) = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
  *reinterpret_cast<float4*>(o + 4) = make_float4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
}

// Layer 0's parameter gradient [27 taps + bias][32] = sum over the rows of (patch / 255 | 1)^T x dy, per camera, on the EXACT
// fp32 matrix pipe (v_mfma_f32_32x32x2_f32: D[32 x 32] += A[32 x 2] B[2 x 32]): one MFMA takes TWO rows -- lane l supplies
// A[i = l & 31][k = l >> 5] = patch value i of row k (a byte through the LUT; i = 27 is the ones column, i > 27 zero) and
// B[k][j = l & 31] = dy[row k][j] -- so there is no LDS staging and no operand split at all.  A wave walks its share of the
// rows; the four waves of a workgroup add their 32 x 32 accumulators through LDS in wave order; per-workgroup partials are added
// in workgroup order by small_reduce_chunks_kernel.  (The first round-4 version staged 32 rows at a time through LDS for fp32
// FMAs: LDS-bound, 296 us + 60 us for a serial 256-slab reduce_slabs.)
typedef float f32x16s __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void small_conv0_wgrad_kernel(const uint8_t* frames, const float* dy, float* part, long rows_cam,
                                                               int n_per_cam, long frame_cam_stride, int hi, int wi, int ho, int wo,
                                                               int chunks) {
  __shared__ float lut[256];
  __shared__ float red[3][32 * 32];
  const int cam = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lut[tid] = (float)tid / 255.0f;
  __syncthreads();
  const long per = ((rows_cam + chunks - 1) / chunks + 7) & ~7L;       // rows per workgroup, a multiple of 8 (2 per MFMA x 4 waves)
  const long r0 = (long)chunk * per, r1 = min(rows_cam, r0 + per);
  const int i = lane & 31, kk = lane >> 5;
  const int poff = i < 27 ? (i / 9) * wi * 3 + (i % 9) : 0;            // byte offset of patch value i inside the patch
  f32x16s acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // this lane's rows: r0 + 2 * wave + kk, + 8, + 16, ...; (image, oy, ox) are carried along instead of divided out per row,
  // and eight rows' loads are in flight before their eight MFMAs
  long m = r0 + 2 * wave + kk;
  const long hw = (long)ho * wo;
  long n = min(m, rows_cam - 1) / hw;
  int rem = (int)(min(m, rows_cam - 1) - n * hw), oy = rem / wo, ox = rem - oy * wo;
  const uint8_t* img = frames + (long)cam * frame_cam_stride * hi * wi * 3 + poff;
  const float* dyc = dy + (long)cam * rows_cam * 32 + i;
  while (m - kk < r1) {   // (wave-uniform: both half-waves advance together)
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = m < r1;
      const long pix = ((n * hi + 2 * oy) * (long)wi + 2 * ox) * 3;
      a[u] = (ok && i < 27) ? lut[img[ok ? pix : 0]] : ((ok && i == 27) ? 1.0f : 0.f);
      b[u] = ok ? dyc[m * 32] : 0.f;
      m += 8; ox += 8;
      if (ox >= wo) { ox -= wo; if (++oy >= ho) { oy = 0; ++n; } }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  // C layout: acc[r] = D[row (r & 3) + 8 (r >> 2) + 4 kk][col i]; waves 1..3 hand theirs over, wave 0 adds them in wave order
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][((r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + i] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    float* o = part + ((long)cam * chunks + chunk) * 28 * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
      const float v = ((acc[r] + red[0][row * 32 + i]) + red[1][row * 32 + i]) + red[2][row * 32 + i];
      if (row < 28) o[row * 32 + i] = v;
    }
  }
}

// out[cam][e] = sum over the S per-workgroup partials part[cam][s][e], in a fixed order: sixteen waves take the partials
// s = w, w + 16, ... each, then wave 0 adds the sixteen sums in wave order (deterministic; the generic reduce_slabs walks all S
// one after the other: 60 us for 256 partials)
__global__ __launch_bounds__(1024) void small_reduce_chunks_kernel(const float* part, int S, int n_elem, float* out, long out_cam_stride) {
  __shared__ float red[16][64];
  const int cam = blockIdx.y, l = threadIdx.x & 63, e = blockIdx.x * 64 + l, w = threadIdx.x >> 6;
  float s = 0.f;
  if (e < n_elem)
    for (int k = w; k < S; k += 16) s += part[((long)cam * S + k) * n_elem + e];
  red[w][l] = s;
  __syncthreads();
  if (w == 0 && e < n_elem) {
    float t = red[0][l];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][l];
    out[(long)cam * out_cam_stride + e] = t;
  }
}

// (An exact-fp32-MFMA weight-gradient kernel for layers 1-3 -- operands straight from global memory, the scheme of
// small_conv0_wgrad_kernel -- was measured slower than the bf16x3 GEMM below in round 4 (layer 1: +57 us, layer 2: +235 us,
// layer 3: +168 us, profiles/r04_ab_small_wgrad.log) and removed in round 5.)

// tab[m] = offset (floats) of the first element of im2col row m = (image, oy, ox) in the layer's NHWC input
__global__ __launch_bounds__(256) void small_patch_table_kernel(int* tab, long rows, int hi, int wi, int ho, int wo, int cin) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= rows) return;
  const long n = m / ((long)ho * wo);
  const int rem = (int)(m - n * (long)ho * wo), oy = rem / wo, ox = rem - oy * wo;
  tab[m] = (int)(((n * hi + 2 * oy) * wi + 2 * ox) * cin);
}

// pooled[cam][img][c] = mean over the P pixels of act[(cam*n + img)*P + p][c]   (jnp.mean(x, axis=(-3, -2)))
__global__ __launch_bounds__(256) void small_avgpool_kernel(const float* act, float* pooled, int n, int P, long pooled_cam_stride) {
  const int img = blockIdx.x, c = threadIdx.x;   // 256 channels
  const int cam = img / n, i = img - cam * n;
  const float* a = act + (long)img * P * 256 + c;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += a[(long)p * 256];
  pooled[(long)cam * pooled_cam_stride + (long)i * 256 + c] = s / (float)P;
}

// dact[(img*P + p)][c] = dpooled[cam][i][c] / P where act > 0
__global__ __launch_bounds__(256) void small_avgpool_bwd_kernel(const float* dpooled, const float* act, float* dact, int n, int P,
                                                               long dp_cam_stride) {
  const int img = blockIdx.x, c = threadIdx.x;
  const int cam = img / n, i = img - cam * n;
  const float g = dpooled[(long)cam * dp_cam_stride + (long)i * 256 + c] / (float)P;
  const long base = (long)img * P * 256 + c;
  for (int p = 0; p < P; ++p) dact[base + (long)p * 256] = act[base + (long)p * 256] > 0.f ? g : 0.f;
}

// col2im as a gather, fused with the ReLU mask of the layer below:
// dx[n][iy][ix][ci] = (x > 0) * sum over taps (ky,kx) with iy-ky = 2*oy, ix-kx = 2*ox in range of dcol[(n,oy,ox)][(ky*3+kx)*cin + ci]
__global__ __launch_bounds__(256) void small_col2im_kernel(const float* dcol, const float* x, float* dx, long n_img, int hi, int wi,
                                                          int ho, int wo, int cin, int pitch) {
  const unsigned e = blockIdx.x * 256u + threadIdx.x;   // (the launcher checks that the element count fits 31 bits)
  const unsigned c4n = cin / 4;
  if (e >= (unsigned)(n_img * hi * wi) * c4n) return;
  const int c4 = (int)(e % c4n);
  unsigned t = e / c4n;
  const int ix = (int)(t % (unsigned)wi);
  t /= (unsigned)wi;
  const int iy = (int)(t % (unsigned)hi);
  const long n = t / (unsigned)hi;
  // Stride 2: a pixel receives from the taps ky = iy mod 2 (+ 2 if that is 0), likewise kx: at most 2 x 2 taps.  All four are
  // fetched UNCONDITIONALLY from clamped positions and the ones that do not exist are dropped afterwards (same order of
  // additions as the loop over all nine taps): with a `continue` in front of every load the ten loads of a thread were ten
  // dependent round trips (125 us per launch on average; profiles/README.md round 4).
  const float4 a = reinterpret_cast<const float4*>(x)[e];
  float4 v[2][2];
  bool ok[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int ky = (iy & 1) + 2 * p, y2 = iy - ky, oy = y2 >> 1;
    const bool oky = ky <= 2 && y2 >= 0 && oy < ho;
    const int oyc = min(max(oy, 0), ho - 1), kyc = min(ky, 2);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kx = (ix & 1) + 2 * q, x2 = ix - kx, ox = x2 >> 1;
      ok[p][q] = oky && kx <= 2 && x2 >= 0 && ox < wo;
      const int oxc = min(max(ox, 0), wo - 1), kxc = min(kx, 2);
      const long m = (n * ho + oyc) * wo + oxc;
      v[p][q] = *reinterpret_cast<const float4*>(dcol + m * pitch + (kyc * 3 + kxc) * cin + c4 * 4);
    }
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (ok[p][q]) { s.x += v[p][q].x; s.y += v[p][q].y; s.z += v[p][q].z; s.w += v[p][q].w; }
  s.x = a.x > 0.f ? s.x : 0.f; s.y = a.y > 0.f ? s.y : 0.f; s.z = a.z > 0.f ? s.z : 0.f; s.w = a.w > 0.f ? s.w : 0.f;
  reinterpret_cast<float4*>(dx)[e] = s;
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
int small_forward(SmallWorkspace& ws, const float* P, long conv_off, long cam_stride, const uint8_t* frames,
                  long frame_cam_stride, int n_cam, int n, float* pooled, long pooled_cam_stride, hipStream_t stream) {
  const long n_img = (long)n_cam * n;
  SERL_REQUIRE(n_img > 0 && n_img <= ws.max_images, "small_forward: %ld images exceed the workspace (%d)", n_img, ws.max_images);
  const SmallDims& d = ws.d;
  SERL_REQUIRE(d.h[kSmallLayers] >= 1 && d.w[kSmallLayers] >= 1, "image too small for the SmallEncoder");
  ProfScope prof("small_encoder_fwd", stream);
  if (!ws.tab_ready) {   // (static shapes: once per workspace)
    for (int l = 1; l < kSmallLayers; ++l) {
      const long r = rows_of(d, l, ws.max_images);
      SERL_REQUIRE((long)ws.max_images * d.h[l] * d.w[l] * kSmallFeat[l] < (1L << 31), "SmallEncoder activation too large for 32-bit offsets");
      hipLaunchKernelGGL(small_patch_table_kernel, dim3(cdiv(r, 256)), dim3(256), 0, stream, ws.tab[l], r, d.h[l], d.w[l],
                         d.h[l + 1], d.w[l + 1], kSmallFeat[l]);
      SERL_HIP(hipGetLastError());
    }
    // one-time: later passes may come in on OTHER streams (sample_actions, a pipelined schedule) and must find the tables complete
    SERL_HIP(hipStreamSynchronize(stream));
    ws.tab_ready = true;
  }
  for (int l = 0; l < kSmallLayers; ++l) {
    const int cin = kSmallFeat[l], cout = kSmallFeat[l + 1], K = 9 * cin + 1, pitch = ldk(l);
    const long rows = rows_of(d, l, n_img), rows_cam = rows_of(d, l, n);
    GemmDesc g{};   // act[cam] = relu(col[cam] ([rows_cam][K+1]) x [kernel ; bias]_cam ([K+1][cout]))
    if (l == 0) {   // directly on the u8 frames, fp32 FMAs (no im2col matrix, no GEMM)
      SERL_REQUIRE(n_cam <= kSmallMaxCams, "SmallEncoder: at most %d cameras", kSmallMaxCams);
      hipLaunchKernelGGL(small_conv0_fwd_kernel, dim3(cdiv(rows_cam * 4, 256), n_cam), dim3(256), 0, stream, frames,
                         P + conv_off, cam_stride, ws.act[0], rows_cam, n, frame_cam_stride, d.h[0], d.w[0], d.h[1], d.w[1]);
      SERL_HIP(hipGetLastError());
      ws.last_frames = frames; ws.last_frame_cam_stride = frame_cam_stride;
      (void)rows; (void)pitch;
      continue;
    } else {
      g.A = ws.act[l - 1]; g.sAm = 0; g.sAk = 1; g.sAb = 0;
      g.gtab = ws.tab[l]; g.gseg = 3 * cin; g.gkbias = 9 * cin; g.gpitch = (long)d.w[l] * cin;
    }
    g.B = P + conv_off + small_conv_offset(l); g.sBk = cout; g.sBn = 1; g.sBb = cam_stride;
    g.C = ws.act[l]; g.ldc = cout; g.sCz = rows_cam * cout;
    g.M = (int)rows_cam; g.N = cout; g.K = K; g.nbatch = n_cam; g.splitk = 1; g.relu = 1;
    int rc = gemm_f32(g, stream);
    if (rc) return rc;
  }
  const int P4 = d.h[kSmallLayers] * d.w[kSmallLayers];
  hipLaunchKernelGGL(small_avgpool_kernel, dim3((int)n_img), dim3(256), 0, stream, ws.act[kSmallLayers - 1], pooled, n, P4,
                     pooled_cam_stride);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int small_backward(SmallWorkspace& ws, const float* P, long conv_off, long cam_stride, int n_cam, int n, const float* dpooled,
                   long dp_cam_stride, float* G, hipStream_t stream) {
  const long n_img = (long)n_cam * n;
  const SmallDims& d = ws.d;
  ProfScope prof("small_encoder_bwd", stream);
  const int P4 = d.h[kSmallLayers] * d.w[kSmallLayers];
  float* dy = ws.dact;
  float* dnext = ws.dact2;
  hipLaunchKernelGGL(small_avgpool_bwd_kernel, dim3((int)n_img), dim3(256), 0, stream, dpooled, ws.act[kSmallLayers - 1], dy, n,
                     P4, dp_cam_stride);
  SERL_HIP(hipGetLastError());
  for (int l = kSmallLayers - 1; l >= 0; --l) {
    const int cin = kSmallFeat[l], cout = kSmallFeat[l + 1], K = 9 * cin + 1, pitch = ldk(l);
    const long rows_cam = rows_of(d, l, n);
    if (l == 0) {   // straight from the u8 frames of the forward pass: per-chunk partial sums, added in chunk order
      SERL_REQUIRE(ws.last_frames != nullptr, "small_backward without a forward pass");
      const int chunks = (int)std::min<long>(kConv0Chunks, std::max<long>(1, rows_cam / 256));   // one workgroup per chunk (latency-bound: many)
      hipLaunchKernelGGL(small_conv0_wgrad_kernel, dim3(chunks, n_cam), dim3(256), 0, stream, ws.last_frames, dy, ws.slabs, rows_cam, n,
                         ws.last_frame_cam_stride, d.h[0], d.w[0], d.h[1], d.w[1], chunks);
      SERL_HIP(hipGetLastError());
      hipLaunchKernelGGL(small_reduce_chunks_kernel, dim3(cdiv(28 * 32, 64), n_cam), dim3(1024), 0, stream, ws.slabs, chunks, 28 * 32,
                         G + conv_off, cam_stride);
      SERL_HIP(hipGetLastError());
      break;   // the pixels need no gradient
    }
    {  // [dkernel ; dbias]_cam = col_cam^T x dy_cam, K-split over the rows, written into the gradient arena
      // (deeper splits -- 512 or 256 rows per workgroup instead of 2048 -- were measured neutral in round 4: 5.84 / 5.87 / 5.89 ms)
      int S = (int)std::min<long>(64, std::max<long>(1, rows_cam / 2048));
      while (S > 1 && (long)S * n_cam * K * cout > ws.slabs_cap) S >>= 1;
      GemmDesc g{};
      {   // col_l^T gathered from the layer's input activations (still in the workspace)
        g.A = ws.act[l - 1]; g.sAm = 1; g.sAk = 0; g.sAb = 0;
        g.gtab = ws.tab[l]; g.gseg = 3 * cin; g.gkbias = 9 * cin; g.gpitch = (long)d.w[l] * cin;
      }
      g.B = dy; g.sBk = cout; g.sBn = 1; g.sBb = rows_cam * cout;
      g.M = K; g.N = cout; g.K = (int)rows_cam; g.nbatch = n_cam; g.splitk = S;
      float* out = G + conv_off + small_conv_offset(l);
      if (S == 1) {
        g.C = out; g.ldc = cout; g.sCz = cam_stride;
        int rc = gemm_f32(g, stream);
        if (rc) return rc;
      } else {
        g.C = ws.slabs; g.ldc = cout; g.sCz = (long)K * cout;
        int rc = gemm_f32(g, stream);
        if (rc) return rc;
        rc = reduce_slabs(ws.slabs, S, (long)K * cout, n_cam, K, cout, nullptr, 0, out, cout, cam_stride, false, stream);
        if (rc) return rc;
      }
    }
    if (l == 0) break;   // the pixels need no gradient
    {  // dcol_cam = dy_cam x kernel_cam^T  (the ones column has no input below it)
      GemmDesc g{};
      g.A = dy; g.sAm = cout; g.sAk = 1; g.sAb = rows_cam * cout;
      g.B = P + conv_off + small_conv_offset(l); g.sBk = 1; g.sBn = cout; g.sBb = cam_stride;
      g.C = ws.dcol; g.ldc = pitch; g.sCz = rows_cam * pitch;
      g.M = (int)rows_cam; g.N = K - 1; g.K = cout; g.nbatch = n_cam; g.splitk = 1;
      int rc = gemm_f32(g, stream);
      if (rc) return rc;
    }
    const long tot = n_img * d.h[l] * d.w[l] * (cin / 4);
    SERL_REQUIRE(tot < (1L << 31), "SmallEncoder activation too large for 32-bit element indices");
    hipLaunchKernelGGL(small_col2im_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, ws.dcol, ws.act[l - 1], dnext, n_img,
                       d.h[l], d.w[l], d.h[l + 1], d.w[l + 1], cin, pitch);
    SERL_HIP(hipGetLastError());
    std::swap(dy, dnext);
  }
  return SERL_OK;
}

}  // namespace serl
