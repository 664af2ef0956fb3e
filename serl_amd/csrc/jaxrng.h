// JAX's PRNG primitives, host + device inline (see jaxrng.hip for sources and what is pinned where): threefry2x32-20, the element
// of jax.random.bits(key, (n,)) at a flat index, bits -> uniform, bits -> normal (XLA's float32 erf_inv).  Included by jaxrng.hip
// (host entry points, the fill kernel) and by heads.hip (Dropout masks / policy noise drawn INSIDE the kernels that consume them).
#pragma once
#include <cmath>
#include <cstdint>

#include <hip/hip_runtime.h>

namespace serl {

// Threefry-2x32, 20 rounds (Random123), counter (x0, x1) under key (k0, k1)
__host__ __device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
#define SERL_TF_ROUND(R) { x0 += x1; x1 = (x1 << (R)) | (x1 >> (32 - (R))); x1 ^= x0; }
#define SERL_TF_A SERL_TF_ROUND(13) SERL_TF_ROUND(15) SERL_TF_ROUND(26) SERL_TF_ROUND(6)
#define SERL_TF_B SERL_TF_ROUND(17) SERL_TF_ROUND(29) SERL_TF_ROUND(16) SERL_TF_ROUND(24)
  SERL_TF_A x0 += ks[1]; x1 += ks[2] + 1u;
  SERL_TF_B x0 += ks[2]; x1 += ks[0] + 2u;
  SERL_TF_A x0 += ks[0]; x1 += ks[1] + 3u;
  SERL_TF_B x0 += ks[1]; x1 += ks[2] + 4u;
  SERL_TF_A x0 += ks[2]; x1 += ks[0] + 5u;
#undef SERL_TF_A
#undef SERL_TF_B
#undef SERL_TF_ROUND
}

// element e of jax's threefry_2x32(key, arange(n)): the counter array is hashed in two halves (an odd n is padded with one
// zero), element e < h pairs with e + h:  out[e] = y0(e, e + h),  out[e + h] = y1(e, e + h),  h = ceil(n / 2)
__host__ __device__ __forceinline__ uint32_t random_bits_at(uint32_t k0, uint32_t k1, uint64_t n, uint64_t e) {
  const uint64_t h = (n + 1) >> 1;
  const bool second = e >= h;
  const uint64_t lo = second ? e - h : e;
  uint32_t x0 = (uint32_t)lo, x1 = (lo + h < n) ? (uint32_t)(lo + h) : 0u;   // (the pad element's counter value is 0)
  threefry2x32(k0, k1, x0, x1);
  return second ? x1 : x0;
}

// bits -> float32 uniform in [0, 1): 23 mantissa bits
__host__ __device__ __forceinline__ float bits_to_unit(uint32_t b) {
  const uint32_t u = (b >> 9) | 0x3F800000u;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f - 1.0f;
}

// XLA's ErfInv32 (Giles' single-precision polynomial)
__host__ __device__ __forceinline__ float erfinv32(float x) {
  float w = -log1pf(-x * x);
  const bool lt = w < 5.0f;
  w = lt ? w - 2.5f : sqrtf(w) - 3.0f;
  float p = lt ? 2.81022636e-08f : -0.000200214257f;
  p = (lt ? 3.43273939e-07f : 0.000100950558f) + p * w;
  p = (lt ? -3.5233877e-06f : 0.00134934322f) + p * w;
  p = (lt ? -4.39150654e-06f : -0.00367342844f) + p * w;
  p = (lt ? 0.00021858087f : 0.00573950773f) + p * w;
  p = (lt ? -0.00125372503f : -0.0076224613f) + p * w;
  p = (lt ? -0.00417768164f : 0.00943887047f) + p * w;
  p = (lt ? 0.246640727f : 1.00167406f) + p * w;
  p = (lt ? 1.50140941f : 2.83297682f) + p * w;
  return fabsf(x) == 1.0f ? x * INFINITY : p * x;
}

// jax.random.normal's element: uniform in [nextafter(-1, 0), 1) then sqrt(2) * erf_inv
__host__ __device__ __forceinline__ float normal_from_bits(uint32_t b) {
  const float lo = -0.99999994f;                       // nextafter(-1, 0)
  const float scale = 1.0f - lo;                       // rounds to 2.0f in float32, as in jax
  float u = bits_to_unit(b) * scale + lo;
  u = fmaxf(lo, u);
  return 1.41421354f * erfinv32(u);                    // np.float32(np.sqrt(2))
}


}  // namespace serl
