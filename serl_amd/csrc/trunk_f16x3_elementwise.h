// Part of the split-fp16 trunk (trunk_f16x3.hip includes these in order; round 6 split the 2,600-line file by kernel family):
// GroupNorm statistics of odd shapes, weight packing (pack_weights / pack_dma_order), the elementwise producers of the split8 layout.
#pragma once
#include "trunk_f16x3_conv_init.h"

namespace serl {

// GroupNorm statistics of a raw conv output, one workgroup per (image, group): shapes whose statistics cannot ride in the
// conv's epilogue (pmode 3).
__global__ void gn_stats_kernel_b(const float* x, double* stats, int P, int Cc) {
  const int n = blockIdx.x / kGnGroups, g = blockIdx.x % kGnGroups;
  const int gs = Cc / kGnGroups;
  const size_t base = (size_t)n * P * Cc + g * gs;
  double s = 0.0, q = 0.0;
  for (int e = threadIdx.x; e < P * gs; e += 256) {
    const int p = e / gs, c = e - p * gs;
    const size_t at = base + (size_t)p * Cc + c;
    const float v = x[at];
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

// w [K][Cout] fp32 -> hi / lo' fp16 [Cout][K] of w * s_n and inv[n] = 1 / s_n.  One workgroup per output channel.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, uint16_t* hi, uint16_t* lo, float* inv, int K, int Cout) {
  __shared__ float red[4];
  const int n = blockIdx.x;
  float m = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, fabsf(w[(size_t)k * Cout + n]));
  const float sc = channel_scale(m, 8, red);
  if (threadIdx.x == 0) inv[n] = 1.0f / sc;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = w[(size_t)k * Cout + n] * sc;
    const _Float16 h = (_Float16)clamp_h(v);
    const _Float16 l = (_Float16)((v - (float)h) * kLoScale);
    hi[(size_t)n * K + k] = __builtin_bit_cast(uint16_t, h);
    lo[(size_t)n * K + k] = __builtin_bit_cast(uint16_t, l);
  }
}

// Copy of the packed planes in the LDS-DMA kernel's piece order: per (64-row block, 16-wide K slot) one contiguous 4 KB block
// [row][position], position p of row r holding unit p ^ ((r >> 2) & 3); unit u = plane (u >> 1), k-half (u & 1).
__global__ __launch_bounds__(256) void pack_dma_order_kernel(const uint16_t* hi, const uint16_t* lo, uint16_t* dma, int K, int Cout) {
  const int nsl = K >> 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per 16-byte unit
  if (e >= (long)2 * Cout * K / 8) return;
  const int t = (int)(e & 255), r64 = t >> 2, pos = t & 3;
  const long blk = e >> 8;
  const int slot = (int)(blk % nsl), j = (int)(blk / nsl);
  const int u = pos ^ ((r64 >> 2) & 3);
  const uint16_t* src = ((u >> 1) ? lo : hi) + (size_t)(j * 64 + r64) * K + slot * 16 + ((u & 1) << 3);
  reinterpret_cast<uint4*>(dma)[e] = *reinterpret_cast<const uint4*>(src);
}

int pack_dma_order_f16x3(const uint16_t* hi, const uint16_t* lo, uint16_t* dma, int K, int Cout, hipStream_t stream) {
  SERL_REQUIRE(K % 32 == 0 && Cout % 64 == 0, "DMA order needs K %% 32 == 0 and Cout %% 64 == 0");
  const long units = (long)2 * Cout * K / 8;
  hipLaunchKernelGGL(pack_dma_order_kernel, dim3(cdiv(units, 256)), dim3(256), 0, stream, hi, lo, dma, K, Cout);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int pack_conv_weights_f16x3(const float* w, uint16_t* hi, uint16_t* lo, float* inv, int K, int Cout, hipStream_t stream) {
  hipLaunchKernelGGL(pack_weights_kernel, dim3(Cout), dim3(256), 0, stream, w, hi, lo, inv, K, Cout);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// ---------------------------------------------------------------------------------------------
// elementwise producers of the split16 layout
// ---------------------------------------------------------------------------------------------
// The "split8" activation layout (same footprint and pixel addressing as the fp32 NHWC tensor): per 8 channels a
// 16-byte unit of hi x8 fp16 followed by a 16-byte unit of lo' x8 fp16.  A conv loader then moves whole 16-byte units
// (global -> LDS, by register or by LDS-DMA) and each unit IS an MFMA k-block of one plane.  Element e indexes
// (pixel, 4-channel group): its hi half lands at byte e*16 - (e&1)*8, its lo' half 16 bytes further.
__device__ __forceinline__ void store_split8(void* out, long e, float4 v) {
  uint2 hi, lo;
  split4(v, hi, lo);
  uint8_t* p = static_cast<uint8_t*>(out) + e * 16 - (e & 1) * 8;
  *reinterpret_cast<uint2*>(p) = hi;
  *reinterpret_cast<uint2*>(p + 16) = lo;
}
__device__ __forceinline__ float4 load_split8(const void* in, long e) {
  const uint8_t* p = static_cast<const uint8_t*>(in) + e * 16 - (e & 1) * 8;
  const uint2 uh = *reinterpret_cast<const uint2*>(p), ul = *reinterpret_cast<const uint2*>(p + 16);
  const h16x2 h0 = __builtin_bit_cast(h16x2, uh.x), h1 = __builtin_bit_cast(h16x2, uh.y);
  const h16x2 l0 = __builtin_bit_cast(h16x2, ul.x), l1 = __builtin_bit_cast(h16x2, ul.y);
  return make_float4((float)h0[0] + (float)l0[0] * kLoInv, (float)h0[1] + (float)l0[1] * kLoInv,
                     (float)h1[0] + (float)l1[0] * kLoInv, (float)h1[1] + (float)l1[1] * kLoInv);
}

// GN + ReLU + max_pool 3x3/2 SAME -> split16   (resnet_v1.py:257-259)
__global__ __launch_bounds__(256) void gn_relu_maxpool_split_kernel(const float* x, GnRef gn, uint4* out, int N,
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
  store_split8(out, e, m);
}

// Second half of the fused pool: completes the windows that cross a tile edge from the neighbours' first row / column,
// then GroupNorm + ReLU on the extreme and conversion to split16.
__global__ __launch_bounds__(256) void pool_finish_split_kernel(const float* pooled, const float* first_rows,
                                                               const float* first_cols, GnRef gn, uint4* out, int N,
                                                               int Ho, int Wo, int tiles_y, int tiles_x) {
  // one thread = 4 channels x 4 consecutive pooled pixels of a row (the GroupNorm coefficients, derived from the
  // fp64 statistics, are computed once per thread)
  const int Hp = Ho >> 1, Wp = Wo >> 1, Wq = Wp >> 2;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * Hp * Wq * 16) return;
  const int c4 = (int)(e & 15);
  long t = e >> 4;
  const int pq = (int)(t % Wq);
  t /= Wq;
  const int py = (int)(t % Hp);
  const int n = (int)(t / Hp);
  float4 s, h;
  gn_coef4(gn, n, c4 * 4, s, h);
  const float4 gm = *reinterpret_cast<const float4*>(gn.gamma + c4 * 4);
  const float4 sg = make_float4(gm.x < 0.f ? -1.f : 1.f, gm.y < 0.f ? -1.f : 1.f, gm.z < 0.f ? -1.f : 1.f, gm.w < 0.f ? -1.f : 1.f);
  const bool edge_row = (py & 7) == 7 && 2 * py + 2 < Ho;  // window row 2py+2 is the first row of the tile below
  const float* rr = first_rows + (((size_t)n * tiles_y + (edge_row ? (2 * py + 2) / 16 : 0)) * Wo) * 64 + c4 * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int px = 4 * pq + j;
    const size_t o = (((size_t)n * Hp + py) * Wp + px) * 16 + c4;
    float4 m = *reinterpret_cast<const float4*>(pooled + o * 4);
    m.x *= sg.x; m.y *= sg.y; m.z *= sg.z; m.w *= sg.w;  // sign-folded domain: extreme == max
    if (edge_row) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * px + dx;
        if (x < Wo) {
          const float4 v = *reinterpret_cast<const float4*>(rr + (size_t)x * 64);
          m.x = fmaxf(m.x, sg.x * v.x); m.y = fmaxf(m.y, sg.y * v.y); m.z = fmaxf(m.z, sg.z * v.z); m.w = fmaxf(m.w, sg.w * v.w);
        }
      }
    }
    if (j == 3 && (px & 7) == 7 && 2 * px + 2 < Wo) {  // window column 2px+2 is the first column of the tile to the right
      const int tcol = (2 * px + 2) / 16;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * py + dy;
        if (y < Ho) {
          const float4 v = *reinterpret_cast<const float4*>(first_cols + (((size_t)n * Ho + y) * tiles_x + tcol) * 64 + c4 * 4);
          m.x = fmaxf(m.x, sg.x * v.x); m.y = fmaxf(m.y, sg.y * v.y); m.z = fmaxf(m.z, sg.z * v.z); m.w = fmaxf(m.w, sg.w * v.w);
        }
      }
    }
    m.x *= sg.x; m.y *= sg.y; m.z *= sg.z; m.w *= sg.w;  // back to the raw extreme
    m.x = fmaxf(m.x * s.x + h.x, 0.f); m.y = fmaxf(m.y * s.y + h.y, 0.f);
    m.z = fmaxf(m.z * s.z + h.z, 0.f); m.w = fmaxf(m.w * s.w + h.w, 0.f);
    store_split8(out, (long)o, m);
  }
}

// relu(GN(raw)) -> split16: the input of a block's second conv
__global__ __launch_bounds__(256) void gn_relu_split_kernel(const float* raw, GnRef gn, uint4* out, int N, int P,
                                                           int Cc) {
  const int c4n = Cc / 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * P * c4n) return;
  const int c4 = (int)(e % c4n);
  const int n = (int)(e / ((long)P * c4n));
  const float4 v = reinterpret_cast<const float4*>(raw)[e];
  float4 s, h;
  gn_coef4(gn, n, c4 * 4, s, h);
  store_split8(out, e, make_float4(fmaxf(v.x * s.x + h.x, 0.f), fmaxf(v.y * s.y + h.y, 0.f),
                                   fmaxf(v.z * s.z + h.z, 0.f), fmaxf(v.w * s.w + h.w, 0.f)));
}

// block output: relu(GN(raw_b) + residual); residual = x (split16) or GN(raw_proj); out split16 or fp32
__global__ __launch_bounds__(256) void block_out_split_kernel(const float* raw, GnRef gn, const uint4* res_split,
                                                             const float* res_raw, GnRef rgn, uint4* out_split,
                                                             float* out_f32, int N, int P, int Cc) {
  const int c4n = Cc / 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * P * c4n) return;
  const int c4 = (int)(e % c4n);
  const int n = (int)(e / ((long)P * c4n));
  const float4 v = reinterpret_cast<const float4*>(raw)[e];
  float4 s, h;
  gn_coef4(gn, n, c4 * 4, s, h);
  float4 r;
  if (res_raw) {
    r = reinterpret_cast<const float4*>(res_raw)[e];
    float4 s2, h2;
    gn_coef4(rgn, n, c4 * 4, s2, h2);
    r.x = r.x * s2.x + h2.x; r.y = r.y * s2.y + h2.y; r.z = r.z * s2.z + h2.z; r.w = r.w * s2.w + h2.w;
  } else {
    r = load_split8(res_split, e);
  }
  float4 o;
  o.x = fmaxf(r.x + (v.x * s.x + h.x), 0.f);
  o.y = fmaxf(r.y + (v.y * s.y + h.y), 0.f);
  o.z = fmaxf(r.z + (v.z * s.z + h.z), 0.f);
  o.w = fmaxf(r.w + (v.w * s.w + h.w), 0.f);
  if (out_f32) reinterpret_cast<float4*>(out_f32)[e] = o;
  else store_split8(out_split, e, o);
}

}  // namespace serl
