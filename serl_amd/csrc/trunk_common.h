// Device helpers shared by the trunk kernels (trunk.hip: exact fp32 MFMA; trunk_f16x3.hip: split-fp16 MFMA).
#pragma once
#include "internal.h"

namespace serl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// XCD-aware bijective workgroup remap (8 XCDs, block b is dispatched to XCD b % 8): gives every
// XCD a contiguous range of tile ids so tiles that share A rows / weights hit the same L2.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// GroupNorm statistics: 64-lane reduction of per-lane partial (sum, sumsq) in 16-lane channel
// segments (+ the two row halves), then one fp64 atomic per 16-channel segment.
// wait = false: the sums are read by a LATER kernel only (unfused launches, the fused projection's tile, conv_init) -- the atomics are issued
// and not waited for (the end of the kernel completes them): no memory round trip in the workgroup's tail.  wait = true (fused exchanges
// inside the launch): see below.
__device__ __forceinline__ void stats_flush(float s, float q, double* stats_ng /* [G][2] of image */,
                                            int chan, int gsize, bool valid, bool wait = true) {
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  s += __shfl_xor(s, 32);
  q += __shfl_xor(q, 32);
  const int lane = threadIdx.x & 63;
  if (valid && (lane & 15) == 0 && lane < 32) {
    const int g = chan / gsize;
    // SYSTEM scope (sc1): performed at the memory side, past the per-XCD L2s -- the fused GroupNorm epilogues read
    // these sums from workgroups on other XCDs while the kernel is still running.  RETURNING atomics: the old value
    // coming back is the only proof that the read-modify-write has been performed (a no-return atomic leaves vmcnt
    // when the L2 has accepted it), and the epilogue's arrival counter must not overtake the sums.
    const double o0 = __hip_atomic_fetch_add(&stats_ng[2 * g], (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const double o1 = __hip_atomic_fetch_add(&stats_ng[2 * g + 1], (double)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifndef SERL_STATS_NORETURN   // (development switch: the racy no-return variant, to prove the stress test catches it)
    if (wait) asm volatile("" ::"v"(o0), "v"(o1));
#else
    (void)o0; (void)o1;
#endif
  }
}


// ---------------------------------------------------------------------------------------------
// GroupNorm applied by the CONSUMER: every kernel that reads a raw conv output derives the per
// (image, channel) scale/shift on the fly from the producer's (sum, sumsq) statistics:
//   y = x*sc + sh,  sc = gamma*rsqrt(var+eps),  sh = beta - mean*sc,  var = max(0, E[x^2]-E[x]^2)
// (flax nn.GroupNorm fast variance, eps 1e-5; resnet_v1.py:119-126,237).  No coefficient table, no
// extra launch between a conv and its consumer.
// ---------------------------------------------------------------------------------------------
struct GnRef {
  const double* stats;  // [N][4][2] (sum, sumsq) of the producing conv; nullptr = identity
  const float* gamma;   // [C]
  const float* beta;    // [C]
  double inv_count;     // 1 / (P * C/4)
  int gsize;            // channels per group
};

__device__ __forceinline__ void gn_coef4(const GnRef& g, int n, int c, float4& sc, float4& sh) {
  const double* st = g.stats + ((size_t)n * kGnGroups + c / g.gsize) * 2;
  const double mean = st[0] * g.inv_count, m2 = st[1] * g.inv_count;
  const float var = fmaxf((float)(m2 - mean * mean), 0.f);
  const float rstd = rsqrtf(var + 1e-5f), mf = (float)mean;
  const float4 ga = *reinterpret_cast<const float4*>(g.gamma + c);
  const float4 be = *reinterpret_cast<const float4*>(g.beta + c);
  sc = make_float4(ga.x * rstd, ga.y * rstd, ga.z * rstd, ga.w * rstd);
  sh = make_float4(be.x - mf * sc.x, be.y - mf * sc.y, be.z - mf * sc.z, be.w - mf * sc.w);
}

struct ConvArgs {
  const float* in;     // [N][Hi][Wi][Cin]
  const float* w;      // [KH*KW*Cin][Cout]
  float* out;          // [N][Ho][Wo][Cout]
  double* stats;       // [N][4][2] or nullptr
  GnRef in_gn;         // GroupNorm+ReLU of the producing layer applied on load (stats == nullptr: none)
  int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad, padw;
  int M, P, tiles_m, tiles_n;
};

// Every output channel n carries its own power-of-two scale s_n (max_k |w[k][n]| * s_n in [128, 256)): the fp16 planes
// hold w * s_n, so neither a strong filter can overflow fp16 nor a weak one sink into its subnormals, and the conv
// epilogue multiplies the fp32 accumulator by inv[n] = 1 / s_n (exact).
struct PackedConvWeights {
  const uint16_t* hi;  // fp16 [Cout][K]  (K = kh*kw*Cin contiguous) of w * s_n
  const uint16_t* lo;  // fp16 [Cout][K]  (w * s_n - float(hi)) * 2^11
  const float* inv;    // [Cout] 1 / s_n
  const uint16_t* dma = nullptr;   // optional copy in the LDS-DMA kernel's piece order (pack_dma_order_f16x3)
};
// w [K][Cout] fp32 (HWIO) -> hi / lo' fp16 [Cout][K], inv [Cout]
int pack_conv_weights_f16x3(const float* w, uint16_t* hi, uint16_t* lo, float* inv, int K, int Cout, hipStream_t stream);
// hi / lo' planes [Cout][K] -> [Cout/64][K/32 chunks][plane][64 rows][4 swizzled 16-byte slots]: a 16-row LDS-DMA weight piece is
// then 1 KB of consecutive bytes in lane order
int pack_dma_order_f16x3(const uint16_t* hi, const uint16_t* lo, uint16_t* dma, int K, int Cout, hipStream_t stream);
// whole trunk in split-fp16 arithmetic (trunk_f16x3.hip)
int trunk_forward_f16x3(const TrunkWeights& w, TrunkWorkspace& ws, TrunkPacked& pk, const uint8_t* frames, int N,
                        float* feats_out, hipStream_t stream, int stage_begin = -1, int stage_end = kTrunkStages - 1);

// conv_init in split-fp16 (weights re-indexed and padded to [64][176])
int pack_conv_init_f16x3(const float* w, uint16_t* hi, uint16_t* lo, float* inv, hipStream_t stream);
// pool_gamma != nullptr: fused 3x3/2 max-pool (trunk_f16x3.hip); `out` then receives the three compact outputs;
// complete_pool: the pooled tensor is final (whole-image chunks), no pool_finish pass needed
int launch_conv_init_f16x3(const uint8_t* img, PackedConvWeights w, float* out, double* stats, int N, int H, int W,
                           int Ho, int Wo, hipStream_t stream, const float* pool_gamma = nullptr, int* ticket = nullptr,
                           bool complete_pool = false);

}  // namespace serl
