// Dense building blocks of the SAC/DrQ update on MI355X: strided/batched/split-K fp32 MFMA GEMM and
// the fused elementwise/normalisation kernels around it (LayerNorm+tanh forward/backward,
// SpatialLearnedEmbeddings, tanh-Gaussian policy head, REDQ target, losses, 3x Adam + target EMA).
// Reference semantics are cited per kernel (paths relative to serl_launcher/serl_launcher/).
#include <algorithm>
#include <cstdlib>

#include "heads.h"
#include "jaxrng.h"
#include "prof.h"

namespace serl {

// every kernel launch of the update chain is counted (serl_debug_chain_launches: the tests pin the chain's launch count)
long g_chain_launches = 0;
#define SERL_LAUNCH_CHAIN(...) do { ++g_chain_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =============================================================================================
// GEMM  C[z] = A[b] * B[b] over K-split `s`   (z = b*splitk + s)
// 64x64 tile per 256-thread workgroup (2x2 waves, one 32x32 fp32-MFMA tile each), BK = 32.
// The update chain runs beside the frozen trunk of the next batch, whose conv workgroups own nearly the
// whole register file and LDS of every CU; a chain kernel starts only in the space a retiring conv
// workgroup frees.  The kernel is therefore built lean -- ~60 VGPRs, 18 KB LDS (single-buffered, next chunk
// prefetched in registers) -- so that several of its workgroups fit into one freed conv slot.
// Operands are read with 16-byte vectors along whichever dimension is contiguous (transposes are free:
// A may be k- or m-contiguous, B k- or n-contiguous).  A k-contiguous operand is stored [row][k] with a
// 36-float pitch, so one conflict-free ds_read_b128 feeds four MFMAs: inside each block of 8 k's, lanes 0-31
// take k = 0..3 and lanes 32-63 k = 4..7 (the K order of a sum is free).  A row-contiguous operand keeps its
// orientation in LDS, [k][row] with a 68-float pitch: its 16-byte global vectors become single conflict-free
// ds_write_b128 and a fragment is four conflict-free ds_read_b32 (consecutive lanes = consecutive rows).  Round 2
// transposed such operands on the way into LDS with scalar stores at a 36-float row pitch: 8 lanes per bank,
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.62 for gemm_f32_kernel<true, false> (profiles/r02_mfma_counters.json).
// Edges are zero-filled; slabs are summed by the consumer (reduce_slabs / ln_tanh_fwd), which keeps the
// K-split deterministic.  Arbitrary element strides fall back to gemm_f32_strided_kernel.
// =============================================================================================
constexpr int kGBM = 64, kGBN = 64, kGBK = 32, kGP = 36, kGPT = 68;
constexpr int kGTile = kGBM * kGP;   // floats per operand tile in LDS (>= kGBK * kGPT)
static_assert(kGBK * kGPT <= kGTile, "the [k][row] layout must fit the tile");
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // global loads need only 4-byte alignment

// 4 consecutive floats at p of which the first `valid` exist (<= 0: none, >= 4: all); missing ones are 0
__device__ __forceinline__ f32x4 ld4(const float* p, int valid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (valid >= 4) {
    v = *reinterpret_cast<const f32x4u*>(p);
  } else if (valid > 0) {
    v[0] = p[0];
    if (valid > 1) v[1] = p[1];
    if (valid > 2) v[2] = p[2];
  }
  return v;
}

// One operand tile X[r0 .. r0+64)[k0 .. k0+32) with element strides (sR, sK), one of which is 1.
// KFAST: vectors run along k (thread -> row idx>>3, k 4*(idx&7)); otherwise along the rows
// (thread -> k idx>>4, rows 4*(idx&15)) and are transposed on the way into LDS.
template <bool KFAST>
struct OperandLoader {
  const float* p[2];
  int lim[2];   // KFAST: row valid ? 1 : 0      else: number of valid rows from this thread's first row
  int kofs[2];  // k offset of this thread's vector inside the chunk
  long step;    // pointer advance per chunk
  __device__ __forceinline__ void init(const float* X, long sR, long sK, int r0, int R, int k_begin, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      if (KFAST) {
        const int r = idx >> 3;
        kofs[i] = 4 * (idx & 7);
        lim[i] = (r0 + r < R) ? 1 : 0;
        p[i] = X + (long)(r0 + r) * sR + (k_begin + kofs[i]);
      } else {
        const int rq = 4 * (idx & 15);
        kofs[i] = idx >> 4;
        lim[i] = R - (r0 + rq);
        p[i] = X + (long)(k_begin + kofs[i]) * sK + (r0 + rq);
      }
    }
    step = KFAST ? kGBK : kGBK * sK;
  }
  // loads the chunk starting at k0 (elements at k >= k_end are zero) and advances to the next chunk
  __device__ __forceinline__ void load(f32x4 (&r)[2], int k0, int k_end) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int krem = k_end - (k0 + kofs[i]);
      const int valid = KFAST ? (lim[i] ? krem : 0) : (krem > 0 ? lim[i] : 0);
      r[i] = ld4(p[i], valid);
      p[i] += step;
    }
  }
  __device__ __forceinline__ void store(float* S, const f32x4 (&r)[2], int tid) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      if (KFAST) *reinterpret_cast<f32x4*>(S + (idx >> 3) * kGP + 4 * (idx & 7)) = r[i];      // [row][k]
      else *reinterpret_cast<f32x4*>(S + (idx >> 4) * kGPT + 4 * (idx & 15)) = r[i];          // [k][row]
    }
  }
  // the four operand values of this lane for k = 8q + 4*lh + (0..3) of tile row `row`
  __device__ __forceinline__ static f32x4 frag(const float* S, int row, int lh, int q) {
    if (KFAST) return *reinterpret_cast<const f32x4*>(S + row * kGP + 8 * q + 4 * lh);
    const float* p = S + (8 * q + 4 * lh) * kGPT + row;
    return (f32x4){p[0], p[kGPT], p[2 * kGPT], p[3 * kGPT]};
  }
};

// =============================================================================================
// Last-arriver epilogues (GemmDesc::epi).  The update chain is ~60 small dependent kernels; a launch boundary costs it more
// than most of these kernels do.  Where a GEMM was followed by a kernel whose only job was to sum the GEMM's slabs (K-split
// or ensemble members) and apply something row-local (bias + LayerNorm + tanh, the tanh-Gaussian head), the consumer now runs
// inside the GEMM launch, in the workgroup that arrives LAST at the output tile:
//   * every workgroup writes its 64x64 slab tile with 16-byte WRITE-THROUGH stores (buffer_store_dwordx4 sc1: the bytes leave
//     the XCD's L2, so no release fence / L2 write-back is needed), every wave drains its stores (s_waitcnt vmcnt(0), written
//     as inline asm so the compiler cannot drop it), __syncthreads, then ONE lane takes a ticket from the tile's arrival
//     counter with a returning memory-side atomic;
//   * the workgroup that draws the last ticket re-zeroes the counter and reads every slab of the tile with sc1 loads (served
//     past its own L1: no acquire fence / invalidate needed) IN INDEX ORDER -- the same order the separate reduce_slabs /
//     ln_tanh_fwd kernels summed in, so fused and un-fused results are bit-identical (tests assert exact equality).
// This is the "in-launch split-K reduction" recipe of cdna_hip_programming.md (sc1 stores + drain + barrier + relaxed
// ticket; sc1 loads on the reducer): correct for any placement of a tile's workgroups over CUs / XCDs, no spinning (nobody
// waits for anybody), no cache maintenance next to the co-running trunk pass.  The MFMA accumulator layout gives a lane one
// COLUMN of 16 rows; the tile is turned into row-major 16-byte vectors through the (now idle) operand LDS, 16 rows per wave
// and pass.  Slab images are padded scratch (ldc % 64 == 0, whole tiles): no edge handling on the slab side.
// =============================================================================================
constexpr int kEpiPitch = 36;                       // floats per LDS row of the transpose scratch
constexpr int kEpiScratch = 4 * 16 * kEpiPitch;     // floats (4 waves x 16 rows); one more int behind it = the "I am last" flag
constexpr int kSc1 = 16;                            // buffer aux bits: sc1 (write-through store / L1-bypassing load)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, long float_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, kSc1));
}
__device__ __forceinline__ float ld1_sc1(__amdgpu_buffer_rsrc_t r, long float_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(float_off * 4), 0, kSc1));
}

// accumulators of wave (wm, wn) -> slab tile rows m0 + wm*32 .., columns n0 + wn*32 .. as 16-byte sc1 stores
__device__ __forceinline__ void store_tile_sc1(float* C, long ldc, int m0, int n0, const f32x16& acc, float* scratch, int tid) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  float* S = scratch + wave * 16 * kEpiPitch;
  const __amdgpu_buffer_rsrc_t rs = rsrc_of(C);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int r = 8 * p; r < 8 * p + 8; ++r) S[(8 * ((r >> 2) & 1) + 4 * lh + (r & 3)) * kEpiPitch + li] = acc[r];
    __builtin_amdgcn_wave_barrier();   // (wave-private scratch: LDS ops of one wave execute in order)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = (lane >> 3) + 8 * h, c4 = lane & 7;
      const f32x4 v = *reinterpret_cast<const f32x4*>(S + row * kEpiPitch + 4 * c4);
      const long off = (long)(m0 + wm * 32 + 16 * p + row) * ldc + n0 + wn * 32 + 4 * c4;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)(off * 4), 0, kSc1);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// every wave has drained its sc1 stores -> one ticket; true in the workgroup that arrived last (which re-zeroes the counter)
__device__ __forceinline__ bool arrive_is_last(int* ctr, int total, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int last = old == total - 1;
    if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0;
}

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// N(0,1) draw number i of stream `seed` (the element's position in the GLOBAL tensor: the same for whichever rank owns it)
__device__ __forceinline__ float hash_normal(uint64_t seed, uint64_t i) {
  const uint64_t r = mix64(seed ^ mix64(i));
  const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
  const float u2 = (float)(uint32_t)((r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// LayerNorm(eps 1e-6, fast variance) + tanh of ONE row of 256 by one wave (mlp.py:24-31, resnet_v1.py:371-374, encoding.py:66-68):
// pre = bias[g] + sum_s slab[s][row] (slabs read in index order);  y = tanh(gamma[g]*xhat + beta[g]).  LIVE: the slabs were
// written by other workgroups of THIS launch (sc1 loads); otherwise plain loads (the separate ln_tanh_fwd kernel).
template <bool LIVE>
__device__ __forceinline__ void ln_tanh_row256(const LnFwdArgs& a, int row, int lane) {
  constexpr int D = 256;
  const int grp = row / a.rows_per_group;
  const long lrow = row - grp * a.rows_per_group;
  const long base = (long)grp * a.S * a.slab_stride + lrow * D + lane * 4;
  const __amdgpu_buffer_rsrc_t rs = rsrc_of(a.slabs);
  float4 acc4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + (long)grp * a.pstride + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  auto slab = [&](int s) -> f32x4 {
    if (LIVE) return ld_sc1(rs, base + (long)s * a.slab_stride);
    return *reinterpret_cast<const f32x4*>(a.slabs + base + (long)s * a.slab_stride);
  };
  int s = 0;
  for (; s + 4 <= a.S; s += 4) {   // 4 independent slab reads in flight, added in index order
    const f32x4 x0 = slab(s), x1 = slab(s + 1), x2 = slab(s + 2), x3 = slab(s + 3);
    acc4.x += x0[0]; acc4.y += x0[1]; acc4.z += x0[2]; acc4.w += x0[3];
    acc4.x += x1[0]; acc4.y += x1[1]; acc4.z += x1[2]; acc4.w += x1[3];
    acc4.x += x2[0]; acc4.y += x2[1]; acc4.z += x2[2]; acc4.w += x2[3];
    acc4.x += x3[0]; acc4.y += x3[1]; acc4.z += x3[2]; acc4.w += x3[3];
  }
  for (; s < a.S; ++s) {
    const f32x4 x0 = slab(s);
    acc4.x += x0[0]; acc4.y += x0[1]; acc4.z += x0[2]; acc4.w += x0[3];
  }
  const float v[4] = {acc4.x, acc4.y, acc4.z, acc4.w};
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1 += v[j]; s2 += v[j] * v[j]; }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  const float mean = s1 * (1.0f / D), mean2 = s2 * (1.0f / D);
  const float var = fmaxf(mean2 - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-6f);
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = lane * 4 + j;
    const float xh = (v[j] - mean) * rstd;
    const float pre = xh * a.gamma[(long)grp * a.pstride + col] + a.beta[(long)grp * a.pstride + col];
    const float y = a.relu ? fmaxf(pre, 0.f) : tanhf(pre);
    a.y[lrow * a.ld_y + (long)grp * a.y_goff + col] = y;
    if (a.xhat) a.xhat[(long)row * D + col] = xh;
    if (a.dot_out) d += y * a.dot_w[(long)grp * a.dot_gstride + col];
  }
  if (a.rstd && lane == 0) a.rstd[row] = rstd;
  if (a.dot_out) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) d += __shfl_xor(d, off);
    if (lane == 0) a.dot_out[row] = d + a.dot_b[(long)grp * a.dot_b_gstride];
  }
}

// tanh-Gaussian head of rows [r0, r1) from the head GEMM's slabs (actor_critic_nets.py:179-272): one thread per (row, action)
template <bool LIVE>
__device__ __forceinline__ void policy_dist_rows(const PolicyDistArgs& v, int r0, int r1, int tid, int nthreads) {
  const int A = v.A, B = v.B;
  const __amdgpu_buffer_rsrc_t rs = rsrc_of(v.slabs);
  for (int e = r0 * A + tid; e < r1 * A; e += nthreads) {
    const int b = e / A, j = e - b * A;
    float mean = v.bias_mean[j], ls = v.bias_ls[j];
    for (int sp = 0; sp < v.S; ++sp) {   // K-split slabs of the head GEMM: [mean | log_std][split][rows][ld]
      const long o0 = (long)sp * v.slab_stride + (long)b * v.slab_ld + j, o1 = o0 + (long)v.S * v.slab_stride;
      mean += LIVE ? ld1_sc1(rs, o0) : v.slabs[o0];
      ls += LIVE ? ld1_sc1(rs, o1) : v.slabs[o1];
    }
    float ep;
    if (v.eps) ep = v.eps[(long)b * A + j];
    else {
      ep = v.tf ? normal_from_bits(random_bits_at(v.tf_key[0], v.tf_key[1], (uint64_t)v.tf_rows * (uint64_t)A,
                                                 (uint64_t)(v.tf_row0 + b) * (uint64_t)A + (uint64_t)j))
                : hash_normal(v.seed, (uint64_t)(v.row_offset + b) * (uint64_t)A + (uint64_t)j);
      v.eps_out[(long)b * A + j] = ep;
    }
    v.pre[(long)b * A + j] = mean;
    v.pre[((long)B + b) * A + j] = ls;
    const float sd = fminf(fmaxf(expf(ls), v.std_min), v.std_max);
    const float u = mean + sd * ep;
    v.act[(long)b * v.ld_act + j] = tanhf(u);
    v.std_out[(long)b * A + j] = sd;
  }
}
// logp[b] = sum_j(-eps^2/2 - log std - log(2pi)/2) - sum_j 2(log2 - u - softplus(-2u)), in j order (one thread per row)
__device__ __forceinline__ void policy_logp_rows(const PolicyDistArgs& v, int r0, int r1, int tid, int nthreads) {
  const int A = v.A;
  for (int b = r0 + tid; b < r1; b += nthreads) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
      const float e = v.eps ? v.eps[(long)b * A + j] : v.eps_out[(long)b * A + j];
      const float sd = v.std_out[(long)b * A + j];
      const float u = v.pre[(long)b * A + j] + sd * e;
      lp += -0.5f * e * e - logf(sd) - 0.91893853320467274f;
      lp -= 2.f * (0.69314718055994531f - u - softplusf(-2.f * u));
    }
    v.logp[b] = lp;
  }
}

// the epilogue proper; `lds` = the kernel's operand LDS (>= kEpiScratch floats + 1 int), idle after the K loop.
// (A LayerNorm + tanh reducer -- the last arriver of a 64-row tile normalising its rows -- existed in round 4: bit-identical, and
// slower in every schedule than the separate LayerNorm launch that spreads the rows over the chip; removed, numbers in
// profiles/README.md.)
__device__ __forceinline__ void gemm_epilogue(const GemmDesc& g, const f32x16& acc, float* C, int z, int batch, int m0, int n0,
                                              float* lds, int tid) {
  store_tile_sc1(C, g.ldc, m0, n0, acc, lds, tid);
  int* flag = reinterpret_cast<int*>(lds + kEpiScratch);
  const int tiles_m = (g.M + kGBM - 1) / kGBM, tiles_n = (g.N + kGBN - 1) / kGBN;
  if (g.epi == kEpiReduce) {
    const int zg = z / g.zred;
    if (!arrive_is_last(g.ctr + ((long)zg * tiles_m + blockIdx.y) * tiles_n + blockIdx.x, g.zred, flag)) return;
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(g.C + (long)zg * g.zred * g.sCz);
    float* out = g.out + (long)zg * g.out_gstride;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = m0 + 16 * p + (tid >> 4), n = n0 + 4 * (tid & 15);
      if (m >= g.M || n >= g.N) continue;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < g.zred; ++s) sum += ld_sc1(rs, (long)s * g.sCz + (long)m * g.ldc + n);
      float* o = out + (long)m * g.ld_out + n;
      if (n + 4 <= g.N) *reinterpret_cast<f32x4u*>(o) = sum;
      else for (int j = 0; j < g.N - n; ++j) o[j] = sum[j];
    }
  } else {   // kEpiPolicy
    if (!arrive_is_last(g.ctr + blockIdx.y, g.nbatch * g.splitk * tiles_n, flag)) return;
    const int r_end = min(m0 + kGBM, g.M);
    policy_dist_rows<true>(g.pd, m0, r_end, tid, 256);
    __syncthreads();   // (this workgroup's own global writes of pre / std / eps, read back below)
    policy_logp_rows(g.pd, m0, r_end, tid, 256);
    if (blockIdx.y == 0 && tid == 0 && g.pd.alpha_out) g.pd.alpha_out[0] = softplusf(g.pd.lam[0]);
  }
}

// Several independent GEMMs in one launch (the chain is latency-bound: one launch per *kind* of work, not per
// instance): group i owns blockIdx.z in [zend[i-1], zend[i]); the x/y grid is the maximum over the groups.
struct GemmMulti {
  GemmDesc d[kMaxGemmGroups];
  int zend[kMaxGemmGroups];
  int n;
};

template <bool A_KFAST, bool B_KFAST>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmMulti mm) {
  static_assert(2 * kGTile >= kEpiScratch + 4, "the epilogue's transpose scratch lives in the operand LDS");
  __shared__ __attribute__((aligned(16))) float ABf[2 * kGTile];
  float* const As = ABf;            // [m][k] (pitch 36) or [k][m] (pitch 68)
  float* const Bs = ABf + kGTile;   // [n][k]             or [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int grp = 0;
  while (grp + 1 < mm.n && (int)blockIdx.z >= mm.zend[grp]) ++grp;
  const GemmDesc& g = mm.d[grp];
  const int z = blockIdx.z - (grp ? mm.zend[grp - 1] : 0), batch = z / g.splitk, split = z - batch * g.splitk;
  const int m0 = blockIdx.y * kGBM, n0 = blockIdx.x * kGBN;
  if (m0 >= g.M || n0 >= g.N) return;
  const int kper = ((g.K + g.splitk - 1) / g.splitk + kGBK - 1) / kGBK * kGBK;
  const int k_begin = split * kper, k_end = min(g.K, k_begin + kper);
  float* C = g.C + (long)z * g.sCz;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int li = lane & 31, lh = lane >> 5;
  if (k_begin < k_end) {
    OperandLoader<A_KFAST> la;
    OperandLoader<B_KFAST> lb;
    la.init(g.A + (long)batch * g.sAb, g.sAm, g.sAk, m0, g.M, k_begin, tid);
    lb.init(g.B + (long)batch * g.sBb, g.sBn, g.sBk, n0, g.N, k_begin, tid);
    f32x4 ra[2], rb[2];
    la.load(ra, k_begin, k_end);
    lb.load(rb, k_begin, k_end);
    for (int k0 = k_begin; k0 < k_end; k0 += kGBK) {
      la.store(As, ra, tid);
      lb.store(Bs, rb, tid);
      __syncthreads();
      la.load(ra, k0 + kGBK, k_end);  // past k_end: all-zero vectors, nothing is dereferenced
      lb.load(rb, k0 + kGBK, k_end);
#pragma unroll
      for (int q = 0; q < kGBK / 8; ++q) {
        const f32x4 a = OperandLoader<A_KFAST>::frag(As, wm * 32 + li, lh, q);
        const f32x4 b = OperandLoader<B_KFAST>::frag(Bs, wn * 32 + li, lh, q);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
      }
      __syncthreads();
    }
  }
  if (g.epi) { gemm_epilogue(g, acc, C, z, batch, m0, n0, As, tid); return; }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int n = n0 + wn * 32 + li;
    if (m < g.M && n < g.N) C[(long)m * g.ldc + n] = acc[r];
  }
}

// =============================================================================================
// The same GEMM on the bf16 matrix pipe: "bf16x3".  Every fp32 operand is split into three bf16 pieces
//     x = x0 + x1 + x2,   x0 = rn16(x), x1 = rn16(x - x0), x2 = rn16(x - x0 - x1)          (split3 below)
// (bf16 keeps fp32's exponent and 8 significant bits: three pieces cover the 24-bit significand) and the product a*b is taken
// as the six piece products with i + j <= 2
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)
// on v_mfma_f32_32x32x16_bf16, accumulated in fp32; the dropped products a1 b2 + a2 b1 + a2 b2 are below 2^-23 |a b| and of
// either sign -- fp32 round-off class.  Unlike the trunk's split-fp16 scheme this needs NO operand scaling (bf16 has the
// range of fp32), so it takes activations, weights and back-propagated gradients (1e-3 .. 1e-9) alike: 6 MFMAs of 32 cycles
// per 32x32x16 block instead of 8 of 64 -- 2.7x less time on the matrix pipe, which the update chain shares with the frozen
// trunk of the next batch on the other stream.  Same descriptor, grid, K-split and epilogue as gemm_f32_kernel (drop-in);
// SERL_GEMM=f32 keeps the exact kernel as the reference arithmetic.
// LDS: per operand three planes [64 rows][32 k] bf16 (64-byte rows, 16-byte slots XOR-swizzled by (row >> 2) & 3:
// conflict-free ds_read_b128 fragments, as in the trunk kernels).  A k-contiguous operand arrives as 16-byte global
// vectors (4 k of one row -> one 8-byte store per plane); a row-contiguous operand as eight coalesced 4-byte loads per
// thread (8 consecutive k of one row, 64 consecutive rows per wave instruction -> one 16-byte store per plane).
// =============================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// byte offset of 16-byte slot `slot` of tile row `row` in a bf16 plane with BK k's per row (BK = 32: 64-byte rows, slots
// XOR-ed with (row >> 2) & 3; BK = 16: 32-byte rows, slots XOR-ed with (row >> 3) & 1) -- conflict-free ds_read_b128
// fragments for the lane groups of MI355X_MICROARCH.md (rows {0-3, 12-15, 20-27} etc. land on 64 distinct banks)
template <int BK>
__device__ __forceinline__ int xswz(int row, int slot) {
  return BK == 32 ? row * 64 + ((slot ^ ((row >> 2) & 3)) << 4) : row * 32 + ((slot ^ ((row >> 3) & 1)) << 4);
}

// two fp32 values -> their three bf16 pieces, packed (first value in the low half).  ROUND-TO-NEAREST pieces (v_cvt_pk_bf16_f32):
// x0 = rn(x), x1 = rn(x - x0), x2 = rn(x - x0 - x1); both differences are exact in fp32 and |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|
// (x2 itself is exact unless x - x0 needs 17 bits, then it is off by < 2^-25 |x|).  Round 3 split by TRUNCATION: remainders twice
// as large and all of x's sign, so the dropped products a1 b2 + a2 b1 reached 2^-20 |a b| and biased every dot product
// towards zero; with rounded pieces they are <= 2^-23 |a b| and sign-symmetric.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3(float v, float w, unsigned& p0, unsigned& p1, unsigned& p2) {
  const f32x2_t x = {v, w};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
  const f32x2_t r1 = x - (f32x2_t){__builtin_bit_cast(float, p0 << 16), __builtin_bit_cast(float, p0 & 0xffff0000u)};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  const f32x2_t r2 = r1 - (f32x2_t){__builtin_bit_cast(float, p1 << 16), __builtin_bit_cast(float, p1 & 0xffff0000u)};
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}

// One operand tile of 64 rows x BK k's per chunk.  KFAST: 16-byte global vectors along k (BK / 16 per thread); otherwise
// BK / 4 coalesced 4-byte loads per thread: tile row tid & 63, k's (BK / 4) * (tid >> 6) + j.
template <bool KFAST, int BK, bool GATHER = false>
struct XLoader {
  static constexpr int NV = KFAST ? BK / 16 : 1;   // global vectors per thread (KFAST)
  static constexpr int NE = BK / 4;                // floats per thread per chunk
  static constexpr int kPlane = kGBM * BK * 2;     // bytes of one bf16 plane
  const float* p[NV];
  int ok[NV];   // this thread's row exists
  const float* safe = nullptr;   // the operand's first element: where loads of non-existent rows / k's are redirected
  long sK;
  int row0, kq;   // !KFAST: tile row of this thread, first k of its group
  // gather mode (GemmDesc::gtab): the operand is a virtual im2col matrix
  const int* tab = nullptr;   // KFAST: unused after init; !KFAST: row offsets, indexed by k
  int gseg = 0, gkbias = 0, gcnt = 0;
  long gpitch = 0;
  bool bias_row = false;      // !KFAST: this thread's tile row is the bias column (all ones)
  int tpre[BK / 4];           // !KFAST: row offsets of the chunk the NEXT load() call fetches (read one call ahead: the pixel loads
  int tlast = 0;              //         depend on them, and a dependent pair inside one call would make the fetch synchronous)
  // KFAST (forward): tile rows are im2col rows, k runs along a patch; !KFAST (weight gradient): tile rows are patch columns
  __device__ __forceinline__ void init_gather(const GemmDesc& g, int batch, int r0, int R, int k_begin, int tid) {
    gseg = g.gseg; gkbias = g.gkbias; gpitch = g.gpitch; sK = 1;
    safe = g.A;
    if (KFAST) {
      tab = g.gtab + (long)batch * g.M;
      const int ky = k_begin / gseg;
      gcnt = (k_begin - ky * gseg) / BK;   // chunks already consumed in the current kernel row
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i, r = idx / (BK / 4);
        ok[i] = r0 + r < R;
        p[i] = g.A + (ok[i] ? tab[r0 + r] : 0) + (long)ky * gpitch + (k_begin - ky * gseg) + 4 * (idx % (BK / 4));
      }
    } else {
      tab = g.gtab + (long)batch * g.K;
      tlast = g.K - 1;
      row0 = tid & 63; kq = NE * (tid >> 6);
      const int kcol = r0 + row0;
      ok[0] = kcol < R;
      bias_row = kcol == gkbias;
      const int ky = min(kcol, gkbias - 1) / gseg;
      p[0] = g.A + (long)ky * gpitch + (min(kcol, gkbias - 1) - ky * gseg);
#pragma unroll
      for (int j = 0; j < NE; ++j) tpre[j] = tab[min(k_begin + kq + j, tlast)];   // row offsets of the FIRST chunk
    }
  }
  __device__ __forceinline__ void init(const float* X, long sR, long sK_, int r0, int R, int k_begin, int tid) {
    sK = sK_;
    safe = X;
    if (KFAST) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i, r = idx / (BK / 4);
        ok[i] = r0 + r < R;
        p[i] = X + (long)(r0 + r) * sR + k_begin + 4 * (idx % (BK / 4));
      }
    } else {
      row0 = tid & 63; kq = NE * (tid >> 6);
      ok[0] = r0 + row0 < R;
      p[0] = X + (long)(k_begin + kq) * sK + r0 + row0;
    }
  }
  // the chunk starting at k0 (elements at k >= k_end are zero); advances to the next chunk
  __device__ __forceinline__ void load(float (&r)[NE], int k0, int k_end) {
    // The gather paths are BRANCH-FREE and UNTOUCHED like the plain ones below (round 4: with each load in its own exec-masked
    // region the SmallEncoder's GEMMs waited vmcnt(0) right behind every load): an element that does not exist is fetched from
    // the operand's first element and fixed at LDS-store time by clean().
    if (GATHER && KFAST) {   // gather, forward: a chunk lies inside one kernel row (gseg % BK == 0) or is the bias chunk
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int k = k0 + 4 * ((threadIdx.x + 256 * i) % (BK / 4));
        const f32x4 v = *reinterpret_cast<const f32x4u*>((ok[i] && k < gkbias) ? p[i] : safe);
        r[4 * i] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
        p[i] += BK;
      }
      if (++gcnt == gseg / BK) {   // next kernel row
        gcnt = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) p[i] += gpitch - gseg;
      }
    } else if (GATHER) {     // gather, weight gradient: k runs over the im2col rows; this chunk's row offsets were read one call ago
#pragma unroll
      for (int j = 0; j < NE; ++j) r[j] = p[0][tpre[j]];
#pragma unroll
      for (int j = 0; j < NE; ++j) tpre[j] = tab[min(k0 + BK + kq + j, tlast)];
    } else if (KFAST) {
      // BRANCH-FREE and UNTOUCHED: the load is always issued (a vector that lies outside the operand is redirected to the
      // operand's first element) and the raw registers are left alone until store() zeroes the invalid elements -- two chunks
      // later.  With the loads inside exec-masked branches (round 3), or with the zeroing selects right behind them, hipcc
      // waits vmcnt(0) in the same iteration and the "prefetch" is synchronous: ~1 us per 16-wide chunk, matrix pipe 10 % busy.
      // A partially valid vector (K % 4 != 0) reads up to 12 bytes past its row: inside the next row or the 256-byte padding
      // every arena buffer ends with.
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int krem = k_end - (k0 + 4 * ((threadIdx.x + 256 * i) % (BK / 4)));
        const f32x4 v = *reinterpret_cast<const f32x4u*>((ok[i] && krem > 0) ? p[i] : safe);
        r[4 * i] = v[0]; r[4 * i + 1] = v[1]; r[4 * i + 2] = v[2]; r[4 * i + 3] = v[3];
        p[i] += BK;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) r[j] = *((ok[0] && k0 + kq + j < k_end) ? p[0] + (long)j * sK : safe);
      p[0] += (long)BK * sK;
    }
  }
  // zeroes the elements of chunk k0 that do not exist (rows past the operand, k >= k_end) and sets the gather operand's ones
  __device__ __forceinline__ void clean(float (&r)[NE], int k0, int k_end) const {
    if (GATHER && KFAST) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int k = k0 + 4 * ((threadIdx.x + 256 * i) % (BK / 4));
        const bool live = ok[i] && k < gkbias, one = ok[i] && k == gkbias && k < k_end;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[4 * i + j] = live ? r[4 * i + j] : 0.f;
        r[4 * i] = one ? 1.0f : r[4 * i];
      }
      return;
    }
    if (GATHER) {
#pragma unroll
      for (int j = 0; j < NE; ++j) r[j] = (ok[0] && k0 + kq + j < k_end) ? (bias_row ? 1.0f : r[j]) : 0.f;
      return;
    }
    if (KFAST) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int krem = ok[i] ? k_end - (k0 + 4 * ((threadIdx.x + 256 * i) % (BK / 4))) : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[4 * i + j] = krem > j ? r[4 * i + j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NE; ++j) r[j] = (ok[0] && k0 + kq + j < k_end) ? r[j] : 0.f;
    }
  }
  __device__ __forceinline__ void store(uint8_t* S, const float (&r)[NE], int tid) const {
    if (KFAST) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i, row = idx / (BK / 4), k4 = idx % (BK / 4);
        unsigned q0[2], q1[2], q2[2];
        split3(r[4 * i], r[4 * i + 1], q0[0], q1[0], q2[0]);
        split3(r[4 * i + 2], r[4 * i + 3], q0[1], q1[1], q2[1]);
        const int off = xswz<BK>(row, k4 >> 1) + (k4 & 1) * 8;
        *reinterpret_cast<u32x2*>(S + off) = (u32x2){q0[0], q0[1]};
        *reinterpret_cast<u32x2*>(S + kPlane + off) = (u32x2){q1[0], q1[1]};
        *reinterpret_cast<u32x2*>(S + 2 * kPlane + off) = (u32x2){q2[0], q2[1]};
      }
    } else if (NE == 8) {
      unsigned q0[4], q1[4], q2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(r[2 * j], r[2 * j + 1], q0[j], q1[j], q2[j]);
      const int off = xswz<BK>(row0, kq >> 3);
      *reinterpret_cast<u32x4*>(S + off) = (u32x4){q0[0], q0[1], q0[2], q0[3]};
      *reinterpret_cast<u32x4*>(S + kPlane + off) = (u32x4){q1[0], q1[1], q1[2], q1[3]};
      *reinterpret_cast<u32x4*>(S + 2 * kPlane + off) = (u32x4){q2[0], q2[1], q2[2], q2[3]};
    } else {   // 4 k's: half a slot
      unsigned q0[2], q1[2], q2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) split3(r[2 * j], r[2 * j + 1], q0[j], q1[j], q2[j]);
      const int off = xswz<BK>(row0, kq >> 3) + ((kq >> 2) & 1) * 8;
      *reinterpret_cast<u32x2*>(S + off) = (u32x2){q0[0], q0[1]};
      *reinterpret_cast<u32x2*>(S + kPlane + off) = (u32x2){q1[0], q1[1]};
      *reinterpret_cast<u32x2*>(S + 2 * kPlane + off) = (u32x2){q2[0], q2[1]};
    }
  }
};

// BK = 16: 12 KB of LDS per workgroup -- the update chain's GEMMs run BESIDE the frozen trunk of the next batch, whose
// conv workgroups own 131-152 KB of a CU's 160 KB: a 12 KB workgroup fits next to two row-slab (140 KB) or two LDS-DMA
// (131 KB) conv workgroups instead of waiting for one of them to retire and taking its place.
// GATHER: the SmallEncoder's implicit-GEMM operand (GemmDesc::gtab) is compiled in -- a separate instantiation, because a
// run-time "gather or not" inside the chunk loop makes every loaded value a phi and hipcc then waits right behind each load
template <bool A_KFAST, bool B_KFAST, int BK, bool GATHER = false>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmMulti mm) {
  constexpr int kPlane = kGBM * BK * 2;
  static_assert(6 * kPlane >= (kEpiScratch + 4) * 4, "the epilogue's transpose scratch lives in the operand LDS");
  __shared__ __attribute__((aligned(16))) uint8_t AB[6 * kPlane];
  uint8_t* const As = AB;
  uint8_t* const Bs = AB + 3 * kPlane;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int grp = 0;
  while (grp + 1 < mm.n && (int)blockIdx.z >= mm.zend[grp]) ++grp;
  const GemmDesc& g = mm.d[grp];
  const int z = blockIdx.z - (grp ? mm.zend[grp - 1] : 0), batch = z / g.splitk, split = z - batch * g.splitk;
  const int m0 = blockIdx.y * kGBM, n0 = blockIdx.x * kGBN;
  if (m0 >= g.M || n0 >= g.N) return;
  const int kper = ((g.K + g.splitk - 1) / g.splitk + kGBK - 1) / kGBK * kGBK;   // (same K partition as the fp32 kernel)
  const int k_begin = split * kper, k_end = min(g.K, k_begin + kper);
  float* C = g.C + (long)z * g.sCz;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int li = lane & 31, lh = lane >> 5;
  if (k_begin < k_end) {
    XLoader<A_KFAST, BK, GATHER> la;
    XLoader<B_KFAST, BK> lb;
    if (GATHER) la.init_gather(g, batch, m0, g.M, k_begin, tid);   // (the launcher: every group of a gather launch gathers)
    else la.init(g.A + (long)batch * g.sAb, g.sAm, g.sAk, m0, g.M, k_begin, tid);
    lb.init(g.B + (long)batch * g.sBb, g.sBn, g.sBk, n0, g.N, k_begin, tid);
    // TWO chunks in flight (register sets 0 / 1): a chunk is 4 KB per operand, its MFMAs take ~0.2 us, an L2 / Infinity-Cache
    // round trip next to the trunk pass 1-2 us -- one chunk of look-ahead left the K = 4096 bottleneck GEMM at ~1 us per chunk
    float ra[2][BK / 4], rb[2][BK / 4];
    la.load(ra[0], k_begin, k_end);
    lb.load(rb[0], k_begin, k_end);
    la.load(ra[1], k_begin + BK, k_end);  // past k_end: zeros, nothing is dereferenced
    lb.load(rb[1], k_begin + BK, k_end);
    for (int k0 = k_begin; k0 < k_end; k0 += 2 * BK) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = k0 + h * BK;
        if (kk >= k_end) break;   // (uniform)
        la.clean(ra[h], kk, k_end);
        lb.clean(rb[h], kk, k_end);
        la.store(As, ra[h], tid);
        lb.store(Bs, rb[h], tid);
        __syncthreads();
        la.load(ra[h], kk + 2 * BK, k_end);
        lb.load(rb[h], kk + 2 * BK, k_end);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          const int ao = xswz<BK>(wm * 32 + li, 2 * ks + lh), bo = xswz<BK>(wn * 32 + li, 2 * ks + lh);
          bf16x8 a[3], b[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            a[pl] = *reinterpret_cast<const bf16x8*>(As + pl * kPlane + ao);
            b[pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * kPlane + bo);
          }
          // smallest terms first
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        __syncthreads();
      }
    }
  }
  if (g.epi) { gemm_epilogue(g, acc, C, z, batch, m0, n0, reinterpret_cast<float*>(AB), tid); return; }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int n = n0 + wn * 32 + li;
    if (m < g.M && n < g.N) C[(long)m * g.ldc + n] = g.relu ? fmaxf(acc[r], 0.f) : acc[r];
  }
}

// arbitrary element strides: scalar loads, [k][m] LDS tiles
__global__ __launch_bounds__(256) void gemm_f32_strided_kernel(GemmDesc g) {
  constexpr int BK = 16, P = 68, LD = kGBM * BK / 256;
  __shared__ float As[BK][P];
  __shared__ float Bs[BK][P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z, batch = z / g.splitk, split = z - batch * g.splitk;
  const int m0 = blockIdx.y * kGBM, n0 = blockIdx.x * kGBN;
  const int kper = ((g.K + g.splitk - 1) / g.splitk + kGBK - 1) / kGBK * kGBK;
  const int k_begin = split * kper, k_end = min(g.K, k_begin + kper);
  const float* A = g.A + (long)batch * g.sAb;
  const float* B = g.B + (long)batch * g.sBb;
  float* C = g.C + (long)z * g.sCz;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int li = lane & 31, lh = lane >> 5;
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int x = tid & 63, k = (tid >> 6) + 4 * i, gk = k0 + k;
      As[k][x] = (m0 + x < g.M && gk < k_end) ? A[(long)(m0 + x) * g.sAm + (long)gk * g.sAk] : 0.f;
      Bs[k][x] = (n0 + x < g.N && gk < k_end) ? B[(long)gk * g.sBk + (long)(n0 + x) * g.sBn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * ks + lh][wm * 32 + li], Bs[2 * ks + lh][wn * 32 + li], acc, 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int n = n0 + wn * 32 + li;
    if (m < g.M && n < g.N) C[(long)m * g.ldc + n] = acc[r];
  }
}

int gemm_f32_multi(const GemmDesc* gs, int n, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxGemmGroups, "bad GEMM group count %d", n);
  GemmMulti mm{};
  mm.n = n;
  int gx = 0, gy = 0, z = 0;
  // one kernel instantiation per launch: a layout every group supports (a one-row / one-column operand fits both)
  bool a_k = true, a_m = true, b_k = true, b_n = true;
  for (int i = 0; i < n; ++i) {
    a_k = a_k && gs[i].sAk == 1; a_m = a_m && gs[i].sAm == 1;
    b_k = b_k && gs[i].sBk == 1; b_n = b_n && gs[i].sBn == 1;
  }
  const bool vec = (a_k || a_m) && (b_k || b_n);
  SERL_REQUIRE(vec || n == 1, "mixed operand layouts in a GEMM group");
  bool gather = false;   // (gather / relu descriptors exist only in the bf16x3 kernel)
  for (int i = 0; i < n; ++i) {
    gather = gather || gs[i].gtab != nullptr || gs[i].relu;
    SERL_REQUIRE(!gs[i].gtab || (gs[i].gseg > 0 && gs[i].gseg % 16 == 0 && (gs[i].sAk == 1 || gs[i].sAm == 1)), "bad gather GEMM");
    SERL_REQUIRE(!gs[i].relu || gs[i].splitk == 1, "relu epilogue with a K-split");
  }
  SERL_REQUIRE(!gather || vec, "gather GEMM needs a vector layout");
  for (int i = 0; i < n; ++i) {
    const GemmDesc& g = gs[i];
    SERL_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.nbatch > 0 && g.splitk > 0, "bad GEMM shape");
    mm.d[i] = g;
    gx = std::max(gx, cdiv(g.N, kGBN));
    gy = std::max(gy, cdiv(g.M, kGBM));
    z += g.nbatch * g.splitk;
    mm.zend[i] = z;
  }
  dim3 grid(gx, gy, z);
  // SERL_GEMM=f32: the exact fp32-MFMA kernel (v_mfma_f32_32x32x2_f32) instead of bf16x3 -- A/B timing and parity runs
  static const bool exact = []() { const char* e = getenv("SERL_GEMM"); return e && e[0] == 'f'; }();
  bool any_epi = false;
  for (int i = 0; i < n; ++i) any_epi = any_epi || gs[i].epi != kEpiNone;
  SERL_REQUIRE(!any_epi || vec, "epilogue GEMMs need a vector layout");
  if (vec && (!exact || gather)) {   // BK = 16 (12 KB of LDS); BK = 32 (24 KB) was measured 30 us per step slower next to the trunk pass
    bool any_tab = false;
    for (int i = 0; i < n; ++i) any_tab = any_tab || gs[i].gtab != nullptr;
    for (int i = 0; i < n; ++i) SERL_REQUIRE(!any_tab || gs[i].gtab != nullptr, "a gather launch mixes gathered and plain operands");
    if (any_tab) {
      if (a_k && b_k) SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<true, true, 16, true>), grid, dim3(256), 0, stream, mm);
      else if (a_k) SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<true, false, 16, true>), grid, dim3(256), 0, stream, mm);
      else if (b_k) SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<false, true, 16, true>), grid, dim3(256), 0, stream, mm);
      else SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<false, false, 16, true>), grid, dim3(256), 0, stream, mm);
    } else if (a_k && b_k) SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<true, true, 16>), grid, dim3(256), 0, stream, mm);
    else if (a_k) SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<true, false, 16>), grid, dim3(256), 0, stream, mm);
    else if (b_k) SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<false, true, 16>), grid, dim3(256), 0, stream, mm);
    else SERL_LAUNCH_CHAIN((gemm_bf16x3_kernel<false, false, 16>), grid, dim3(256), 0, stream, mm);
  } else if (vec) {
    if (a_k && b_k) SERL_LAUNCH_CHAIN((gemm_f32_kernel<true, true>), grid, dim3(256), 0, stream, mm);
    else if (a_k) SERL_LAUNCH_CHAIN((gemm_f32_kernel<true, false>), grid, dim3(256), 0, stream, mm);
    else if (b_k) SERL_LAUNCH_CHAIN((gemm_f32_kernel<false, true>), grid, dim3(256), 0, stream, mm);
    else SERL_LAUNCH_CHAIN((gemm_f32_kernel<false, false>), grid, dim3(256), 0, stream, mm);
  } else {
    for (int i = 0; i < n; ++i) {
      const GemmDesc& g = gs[i];
      SERL_LAUNCH_CHAIN(gemm_f32_strided_kernel, dim3(cdiv(g.N, kGBN), cdiv(g.M, kGBM), g.nbatch * g.splitk), dim3(256), 0,
                         stream, g);
    }
  }
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int gemm_f32(const GemmDesc& g, hipStream_t stream) { return gemm_f32_multi(&g, 1, stream); }

// out[g][row][col] (+)= bias[g][col] + sum_s slab[g*S + s][row][col]
__global__ void reduce_slabs_kernel(const float* slabs, int S, long slab_stride, int rows, int N,
                                    const float* bias, long bias_gstride, float* out, long ld_out,
                                    long out_gstride, int accumulate, float scale) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = blockIdx.y;
  if (e >= (long)rows * N) return;
  const int row = (int)(e / N), col = (int)(e - (long)row * N);
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += slabs[((long)grp * S + s) * slab_stride + e];
  v *= scale;
  if (bias) v += bias[(long)grp * bias_gstride + col];
  float* o = out + (long)grp * out_gstride + (long)row * ld_out + col;
  *o = accumulate ? *o + v : v;
}

int reduce_slabs(const float* slabs, int S, long slab_stride, int groups, int rows, int N, const float* bias,
                 long bias_gstride, float* out, long ld_out, long out_gstride, bool accumulate,
                 hipStream_t stream, float scale) {
  dim3 grid(cdiv((long)rows * N, 256), groups);
  SERL_LAUNCH_CHAIN(reduce_slabs_kernel, grid, dim3(256), 0, stream, slabs, S, slab_stride, rows, N, bias,
                     bias_gstride, out, ld_out, out_gstride, accumulate ? 1 : 0, scale);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// =============================================================================================
// LayerNorm(eps 1e-6, fast variance) + tanh, one wave per row.   mlp.py:24-31, resnet_v1.py:371-374,
// encoding.py:66-68.   pre = bias[g] + sum_s slab[s][row];  y = tanh(gamma[g]*xhat + beta[g])
// =============================================================================================
template <int VPL>
__global__ __launch_bounds__(256) void ln_tanh_fwd_kernel(Multi<LnFwdArgs> mv) {
  const LnFwdArgs& a = mv.v[blockIdx.y];  // blockIdx.y = independent instance (variant)
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  if (VPL == 4) { ln_tanh_row256<false>(a, row, lane); return; }   // (the same code the fused GEMM epilogue runs)
  const int grp = row / a.rows_per_group;
  constexpr int D = VPL * 64;
  float v[VPL];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int col = lane * VPL + j;
    float x = a.bias ? a.bias[(long)grp * a.pstride + col] : 0.f;
    for (int s = 0; s < a.S; ++s)
      x += a.slabs[(long)(grp * a.S + s) * a.slab_stride + (long)(row - grp * a.rows_per_group) * D + col];
    v[j] = x;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) { s1 += v[j]; s2 += v[j] * v[j]; }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  const float mean = s1 * (1.0f / D), mean2 = s2 * (1.0f / D);
  const float var = fmaxf(mean2 - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-6f);
  float d = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int col = lane * VPL + j;
    const float xh = (v[j] - mean) * rstd;
    const float pre = xh * a.gamma[(long)grp * a.pstride + col] + a.beta[(long)grp * a.pstride + col];
    const float y = a.relu ? fmaxf(pre, 0.f) : tanhf(pre);
    a.y[(long)(row - grp * a.rows_per_group) * a.ld_y + (long)grp * a.y_goff + col] = y;
    if (a.xhat) a.xhat[(long)row * D + col] = xh;
    if (a.dot_out) d += y * a.dot_w[(long)grp * a.dot_gstride + col];
  }
  if (a.rstd && lane == 0) a.rstd[row] = rstd;
  if (a.dot_out) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) d += __shfl_xor(d, off);
    if (lane == 0) a.dot_out[row] = d + a.dot_b[(long)grp * a.dot_b_gstride];
  }
}

int ln_tanh_fwd_multi(const LnFwdArgs* as, int n, int D, hipStream_t stream) {
  SERL_REQUIRE(D == 256 || D == 64, "LayerNorm width %d unsupported (64 or 256)", D);
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad instance count %d", n);
  Multi<LnFwdArgs> mv{};
  int rows = 0;
  for (int i = 0; i < n; ++i) { mv.v[i] = as[i]; rows = std::max(rows, as[i].rows); }
  dim3 grid(cdiv(rows, 4), n);
  if (D == 256) SERL_LAUNCH_CHAIN(ln_tanh_fwd_kernel<4>, grid, dim3(256), 0, stream, mv);
  else SERL_LAUNCH_CHAIN(ln_tanh_fwd_kernel<1>, grid, dim3(256), 0, stream, mv);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// backward of y = tanh(gamma*xhat + beta), xhat = (x-mean)*rstd:
//   dg = dy*(1-y^2);  dxhat = dg*gamma;  dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat))
template <int VPL>
__global__ __launch_bounds__(256) void ln_tanh_bwd_kernel(LnBwdArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  const int grp = row / a.rows_per_group;
  constexpr int D = VPL * 64;
  float dg[VPL], dxh[VPL], xh[VPL];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int col = lane * VPL + j;
    const long lr = row - grp * a.rows_per_group;
    const float y = a.y[lr * a.ld_y + (long)grp * a.y_goff + col];
    const float dy = a.dq_w ? (a.dq ? a.dq[row] : a.dq_const) * a.dq_w[(long)grp * a.dq_w_gstride + col]
                            : a.dy[lr * a.ld_dy + (long)grp * a.dy_goff + col];
    dg[j] = dy * (1.f - y * y);
    xh[j] = a.xhat[(long)row * D + col];
    dxh[j] = dg[j] * a.gamma[(long)grp * a.pstride + col];
    s1 += dxh[j];
    s2 += dxh[j] * xh[j];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  const float m1 = s1 * (1.0f / D), m2 = s2 * (1.0f / D), rstd = a.rstd[row];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int col = lane * VPL + j;
    a.dx[(long)row * D + col] = rstd * (dxh[j] - m1 - xh[j] * m2);
    a.dg[(long)row * D + col] = dg[j];
  }
}

int ln_tanh_bwd(const LnBwdArgs& a, int D, hipStream_t stream) {
  SERL_REQUIRE(D == 256 || D == 64, "LayerNorm width %d unsupported (64 or 256)", D);
  dim3 grid(cdiv(a.rows, 4));
  if (D == 256) SERL_LAUNCH_CHAIN(ln_tanh_bwd_kernel<4>, grid, dim3(256), 0, stream, a);
  else SERL_LAUNCH_CHAIN(ln_tanh_bwd_kernel<1>, grid, dim3(256), 0, stream, a);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// REDQ target of sample b (sac.py:150-176): y = r + discount * mask * min over the selected (or all) target members
// [- alpha * log pi(a'|s') with backup_entropy, outside the discount as the reference has it]
__device__ __forceinline__ float redq_target(const LossArgs& L, int b) {
  float mq;
  if (L.sel.n > 0) {
    mq = L.qt[(long)L.sel.idx[0] * L.B + b];
    for (int k = 1; k < L.sel.n; ++k) mq = fminf(mq, L.qt[(long)L.sel.idx[k] * L.B + b]);
  } else {
    mq = L.qt[b];
    for (int e = 1; e < L.E; ++e) mq = fminf(mq, L.qt[(long)e * L.B + b]);
  }
  float y = L.reward[b] + L.discount * L.mask[b] * mq;
  if (L.logp_next) y -= L.alpha[0] * L.logp_next[b];
  return y;
}

// critic loss by ONE workgroup of 256 threads (deterministic reductions); the body of critic_loss_kernel
__device__ void critic_loss_body(const LossArgs& L, float (*red)[256]) {
  float s_d2 = 0.f, s_q = 0.f, s_y = 0.f, s_dq = 0.f;
  float s_e[16];  // per-member sums of dQ (per_member: E <= 16)
#pragma unroll
  for (int e = 0; e < 16; ++e) s_e[e] = 0.f;
  for (int b = threadIdx.x; b < L.B; b += 256) {
    const float y = redq_target(L, b);
    L.y_out[b] = y;
    s_y += y;
    for (int e = 0; e < L.E; ++e) {
      const float qv = L.q[(long)e * L.B + b];
      const float d = qv - y;
      const float g = 2.f * d * L.inv_norm;
      L.dq[(long)e * L.B + b] = g;
      s_d2 += d * d;
      s_q += qv;
      s_dq += g;
      if (L.per_member) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k == e) s_e[k] += g;
      }
    }
  }
  red[0][threadIdx.x] = s_d2; red[1][threadIdx.x] = s_q; red[2][threadIdx.x] = s_y; red[3][threadIdx.x] = s_dq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    L.scalars[0] = red[0][0]; L.scalars[1] = red[1][0]; L.scalars[2] = red[2][0];
    if (!L.per_member) *L.dbias = red[3][0];
  }
  if (L.per_member) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < L.E) {  // (uniform)
        __syncthreads();
        red[0][threadIdx.x] = s_e[k];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
          if ((int)threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o];
          __syncthreads();
        }
        if (threadIdx.x == 0) L.dbias[k] = red[0][0];
      }
    }
  }
}

template <int VPL>
__device__ __forceinline__ void ln_tanh_bwd_row(const LnBwdArgs& a, const LossArgs& L, int row, int lane) {
  const int grp = row / a.rows_per_group;
  constexpr int D = VPL * 64;
  float dg[VPL], dxh[VPL], xh[VPL];
  float s1 = 0.f, s2 = 0.f;
  const long lr = row - grp * a.rows_per_group;
  float dqr = 0.f;
  if (a.dq_w) dqr = a.dq_inline ? 2.f * (L.q[row] - redq_target(L, (int)lr)) * L.inv_norm : (a.dq ? a.dq[row] : a.dq_const);
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int col = lane * VPL + j;
    const float y = a.y[lr * a.ld_y + (long)grp * a.y_goff + col];
    const float dy = a.dq_w ? dqr * a.dq_w[(long)grp * a.dq_w_gstride + col] : a.dy[lr * a.ld_dy + (long)grp * a.dy_goff + col];
    dg[j] = dy * (1.f - y * y);
    xh[j] = a.xhat[(long)row * D + col];
    dxh[j] = dg[j] * a.gamma[(long)grp * a.pstride + col];
    s1 += dxh[j];
    s2 += dxh[j] * xh[j];
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  const float m1 = s1 * (1.0f / D), m2 = s2 * (1.0f / D), rstd = a.rstd[row];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int col = lane * VPL + j;
    a.dx[(long)row * D + col] = rstd * (dxh[j] - m1 - xh[j] * m2);
    a.dg[(long)row * D + col] = dg[j];
  }
}

// blockIdx.y = instance (its own width); the LAST workgroup of instance 0 is the critic-loss rider when loss.on
__global__ __launch_bounds__(256) void ln_tanh_bwd_multi_kernel(Multi<LnBwdArgs> mv, LossArgs loss) {
  __shared__ float red[4][256];
  if (loss.on && blockIdx.y == 0 && blockIdx.x == gridDim.x - 1) { critic_loss_body(loss, red); return; }
  const LnBwdArgs& a = mv.v[blockIdx.y];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.rows) return;
  if (a.D == 256) ln_tanh_bwd_row<4>(a, loss, row, lane);
  else ln_tanh_bwd_row<1>(a, loss, row, lane);
}

int ln_tanh_bwd_multi(const LnBwdArgs* as, int n, const LossArgs& loss, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad instance count %d", n);
  Multi<LnBwdArgs> mv{};
  int rows = 0;
  for (int i = 0; i < n; ++i) {
    SERL_REQUIRE(as[i].D == 256 || as[i].D == 64, "LayerNorm width %d unsupported (64 or 256)", as[i].D);
    SERL_REQUIRE(!as[i].dq_inline || (loss.on && as[i].dq_w && as[i].rows == loss.E * loss.B && as[i].rows_per_group == loss.B),
                 "inline dQ needs the loss arguments of the launch");
    mv.v[i] = as[i];
    rows = std::max(rows, as[i].rows);
  }
  SERL_LAUNCH_CHAIN(ln_tanh_bwd_multi_kernel, dim3(cdiv(rows, 4) + (loss.on ? 1 : 0), n), dim3(256), 0, stream, mv, loss);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// out[g][col] (+)= sum_{r in group g} X[r][col] * (Y ? Y[r][col] : 1)     (bias / LN-param grads)
__global__ __launch_bounds__(256) void colsum_kernel(const float* X, const float* Y, int rows_per_group,
                                                    int D, float* out, long out_gstride, int accumulate) {
  __shared__ float red[4][64];
  const int grp = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float s = 0.f;
  if (col < D) {
    const long base = (long)grp * rows_per_group;
    for (int r = part; r < rows_per_group; r += 4) {
      const float x = X[(base + r) * D + col];
      s += Y ? x * Y[(base + r) * D + col] : x;
    }
  }
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && col < D) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    float* o = out + (long)grp * out_gstride + col;
    *o = accumulate ? *o + t : t;
  }
}

int colsum(const float* X, const float* Y, int groups, int rows_per_group, int D, float* out,
           long out_gstride, bool accumulate, hipStream_t stream) {
  SERL_LAUNCH_CHAIN(colsum_kernel, dim3(cdiv(D, 64), groups), dim3(256), 0, stream, X, Y, rows_per_group, D,
                     out, out_gstride, accumulate ? 1 : 0);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// fused parameter gradients of one Dense->LN->tanh layer (one pass over dg, xhat, dpre):
//   dgamma[g][j] = sum_r dg*xhat,  dbeta[g][j] = sum_r dg,  dbias[g][j] = sum_r dpre
struct ColsumMulti { Colsum3Args v[kMaxColsum]; };
__global__ __launch_bounds__(512) void colsum3_kernel(ColsumMulti mv) {
  const Colsum3Args& a = mv.v[blockIdx.z];  // blockIdx.z = layer
  __shared__ float red[3][8][64];
  const int grp = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  if (grp >= a.groups || (int)blockIdx.x * 64 >= a.D) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (a.mode == 2) {   // a vector's sum (loss scalars), in the order policy_dist_fwd_kernel sums log-probs: 256 strided partial
    float* r1 = &red[0][0][0];   // sums, then a halving tree (fused and un-fused chains agree to the bit)
    if (threadIdx.x < 256) {
      for (int r = threadIdx.x; r < a.rows_per_group; r += 256) s1 += a.dg[r];
      r1[threadIdx.x] = s1;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) r1[threadIdx.x] += r1[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) a.o_beta[0] = r1[0];
    return;
  }
  if (col < a.D) {
    const long base = (long)grp * a.rows_per_group;
    if (a.mode == 1) {   // (four row phases combined as colsum_kernel combines them)
      if (part < 4)
        for (int r = part; r < a.rows_per_group; r += 4) s1 += a.dg[(base + r) * a.D + col];
    } else {
      for (int r = part; r < a.rows_per_group; r += 8) {
        const long e = (base + r) * a.D + col;
        const float g = a.dg[e];
        s0 += g * a.xhat[e];
        s1 += g;
        s2 += a.dpre[e];
      }
    }
  }
  red[0][part][threadIdx.x & 63] = s0;
  red[1][part][threadIdx.x & 63] = s1;
  red[2][part][threadIdx.x & 63] = s2;
  __syncthreads();
  if (a.mode == 1) {
    if (part == 1 && col < a.D) {
      const int c = threadIdx.x & 63;
      a.o_beta[(long)grp * a.gstride + col] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
    return;
  }
  if (part < 3 && col < a.D) {
    const int c = threadIdx.x & 63;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[part][k][c];
    float* o = (part == 0 ? a.o_gamma : (part == 1 ? a.o_beta : a.o_bias)) + (long)grp * a.gstride + col;
    *o = t;
  }
}

int colsum3_multi(const Colsum3Args* vs, int n, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxColsum, "bad layer count %d", n);
  ColsumMulti mv{};
  int gx = 0, gy = 0;
  for (int i = 0; i < n; ++i) {
    mv.v[i] = vs[i];
    gx = std::max(gx, cdiv(vs[i].D, 64));
    gy = std::max(gy, vs[i].groups);
  }
  SERL_LAUNCH_CHAIN(colsum3_kernel, dim3(gx, gy, n), dim3(512), 0, stream, mv);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int colsum3(const float* dg, const float* xhat, const float* dpre, int groups, int rows_per_group, int D,
            float* o_gamma, float* o_beta, float* o_bias, long gstride, hipStream_t stream) {
  const Colsum3Args v{dg, xhat, dpre, groups, rows_per_group, D, o_gamma, o_beta, o_bias, gstride, 0};
  return colsum3_multi(&v, 1, stream);
}

// =============================================================================================
// SpatialLearnedEmbeddings (resnet_v1.py:94-111): f[n][c*F+j] = sum_hw x[n][hw][c]*K[hw][c][j]
// (+ Dropout(0.1) keep-mask, resnet_v1.py:351).  F == 8.
// =============================================================================================
__global__ __launch_bounds__(256) void sle_fwd_kernel(Multi<SleFwdArgs> mv, float keep_scale, int N, int HW, int Cc,
                                                     long xs, long ks, long ms, long fs) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)N * Cc) return;
  const SleFwdArgs& v = mv.v[blockIdx.z];  // blockIdx.z = instance, blockIdx.y = camera
  const float* x = v.x + blockIdx.y * xs;
  const float* K = v.K + blockIdx.y * ks;
  float* f = v.f + blockIdx.y * fs;
  const uint8_t* mask = v.mask ? v.mask + blockIdx.y * ms : nullptr;
  const int n = (int)(e / Cc), c = (int)(e - (long)n * Cc);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int hw = 0; hw < HW; ++hw) {
    const float xv = x[((long)n * HW + hw) * Cc + c];
    const float4 k0 = *reinterpret_cast<const float4*>(K + ((long)hw * Cc + c) * 8);
    const float4 k1 = *reinterpret_cast<const float4*>(K + ((long)hw * Cc + c) * 8 + 4);
    acc[0] += xv * k0.x; acc[1] += xv * k0.y; acc[2] += xv * k0.z; acc[3] += xv * k0.w;
    acc[4] += xv * k1.x; acc[5] += xv * k1.y; acc[6] += xv * k1.z; acc[7] += xv * k1.w;
  }
  float* o = f + (long)n * Cc * 8 + (long)c * 8;
  if (mask) {
    const uint8_t* m = mask + (long)n * Cc * 8 + (long)c * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = m[j] ? acc[j] * keep_scale : 0.f;
  }
  *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

int sle_fwd_multi(const SleFwdArgs* vs, int n, float keep_scale, int N, int HW, int Cc, int groups, long x_gs, long k_gs,
                  long mask_gs, long f_gs, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad instance count %d", n);
  Multi<SleFwdArgs> mv{};
  for (int i = 0; i < n; ++i) mv.v[i] = vs[i];
  SERL_LAUNCH_CHAIN(sle_fwd_kernel, dim3(cdiv((long)N * Cc, 256), groups, n), dim3(256), 0, stream, mv, keep_scale, N,
                     HW, Cc, x_gs, k_gs, mask_gs, f_gs);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}


// proprio branch (encoding.py:55-70) of ONE row by one wave, lane = output feature (also the body of proprio_fwd_kernel)
__device__ __forceinline__ void proprio_row(const ProprioArgs& a, int S, int row, int lane) {
  float v = a.b[lane];
  for (int s0 = 0; s0 < S; s0 += 64) {  // the state row travels once, coalesced, and is broadcast lane by lane
    const float mine = (s0 + lane < S) ? a.state[(long)row * S + s0 + lane] : 0.f;
    const int cnt = min(64, S - s0);
    int s = 0;
    for (; s + 8 <= cnt; s += 8) {
      float w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = a.W[(s0 + s + j) * 64 + lane];
#pragma unroll
      for (int j = 0; j < 8; ++j) v += __shfl(mine, s + j) * w[j];
    }
    for (; s < cnt; ++s) v += __shfl(mine, s) * a.W[(s0 + s) * 64 + lane];
  }
  float s1 = v, s2 = v * v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  const float mean = s1 * (1.0f / 64), var = fmaxf(s2 * (1.0f / 64) - mean * mean, 0.f);
  const float rstd = rsqrtf(var + 1e-6f), xh = (v - mean) * rstd;
  a.y[(long)row * a.ld_y + lane] = tanhf(xh * a.gamma[lane] + a.beta[lane]);
  if (a.xhat) a.xhat[(long)row * 64 + lane] = xh;
  if (a.rstd && lane == 0) a.rstd[row] = rstd;
  // optional column copy riding along (the batch's actions into the critic input [enc | action])
  if (a.copy_dst && lane < a.copy_cols) a.copy_dst[(long)row * a.ld_copy_dst + lane] = a.copy_src[(long)row * a.ld_copy_src + lane];
}

// SpatialLearnedEmbeddings, channel-blocked: a workgroup = 256 channels x kSleNb samples.  The embedding kernel K
// ([HW][C][8]: 256 KB per camera at 4x4x512) is read ONCE per workgroup, four pixels at a time into registers, instead of
// once per sample (round 3: 393 MB of L2 reads per update phase, 30 us alone / 160-190 us next to the trunk pass); every
// sample's 8 outputs stay in registers across the pixel loop (same hw-ascending sum as before: bit-identical).  The Dropout(0.1)
// keep-mask (resnet_v1.py:351), when not supplied, is hashed from (seed, camera, GLOBAL sample, channel): three 64-bit hashes
// give the eight 24-bit uniforms of a thread's eight outputs -- no mask tensor, no gen_noise launch.  Workgroups past the SLE
// range run the proprio branch of the same instance (camera 0 only): one launch instead of two.
// kSleNb samples per workgroup; a rank's share of a data-parallel batch (<= 64 samples) takes 2, which keeps 4x more workgroups
// in flight (96 workgroups of 8 samples each took 64 us next to the trunk pass at 32 samples)
template <int kSleNb>
__global__ __launch_bounds__(256) void sle_proprio_fwd_kernel(Multi<SleFwdArgs> mv, Multi<ProprioArgs> pv, int has_proprio,
                                                              float keep_scale, unsigned keep_thr, float keep_p, int N, int HW, int Cc,
                                                              long xs, long ks, long ms, long fs, int S, int nb_sle) {
  if ((int)blockIdx.x >= nb_sle) {
    if (!has_proprio || blockIdx.y != 0) return;
    const int row = ((int)blockIdx.x - nb_sle) * 4 + (threadIdx.x >> 6);
    if (row < N) proprio_row(pv.v[blockIdx.z], S, row, threadIdx.x & 63);
    return;
  }
  const SleFwdArgs& v = mv.v[blockIdx.z];  // blockIdx.z = instance, blockIdx.y = camera
  const int cblocks = (Cc + 255) / 256;
  const int c = ((int)blockIdx.x % cblocks) * 256 + threadIdx.x, n0 = ((int)blockIdx.x / cblocks) * kSleNb;
  if (c >= Cc) return;
  const float* x = v.x + blockIdx.y * xs + (long)n0 * HW * Cc + c;
  const float* K = v.K + blockIdx.y * ks + (long)c * 8;
  const int ns = min(kSleNb, N - n0);
  float acc[kSleNb][8];
#pragma unroll
  for (int s = 0; s < kSleNb; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[s][j] = 0.f;
  for (int hw0 = 0; hw0 < HW; hw0 += 4) {
    float4 k0[4], k1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hw = min(hw0 + i, HW - 1);
      k0[i] = *reinterpret_cast<const float4*>(K + (long)hw * Cc * 8);
      k1[i] = *reinterpret_cast<const float4*>(K + (long)hw * Cc * 8 + 4);
    }
    // (no branch on the sample count: samples past the end re-read the last one and are never stored -- a branch per sample kept
    //  the compiler from issuing the 32 pixel loads of a chunk together: 48 us against 36 for the un-blocked kernel)
    float xv[kSleNb][4];
#pragma unroll
    for (int s = 0; s < kSleNb; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[s][i] = x[((long)min(s, ns - 1) * HW + min(hw0 + i, HW - 1)) * Cc];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (hw0 + i < HW) {   // (uniform; pixels past HW must not touch the sums)
#pragma unroll
        for (int s = 0; s < kSleNb; ++s) {
          acc[s][0] += xv[s][i] * k0[i].x; acc[s][1] += xv[s][i] * k0[i].y; acc[s][2] += xv[s][i] * k0[i].z; acc[s][3] += xv[s][i] * k0[i].w;
          acc[s][4] += xv[s][i] * k1[i].x; acc[s][5] += xv[s][i] * k1[i].y; acc[s][6] += xv[s][i] * k1[i].z; acc[s][7] += xv[s][i] * k1[i].w;
        }
      }
    }
  }
  float* f = v.f + blockIdx.y * fs + (long)n0 * Cc * 8 + (long)c * 8;
  const uint8_t* mask = v.mask ? v.mask + blockIdx.y * ms + (long)n0 * Cc * 8 + (long)c * 8 : nullptr;
#pragma unroll
  for (int s = 0; s < kSleNb; ++s) {
    if (s >= ns) break;
    if (mask) {
      const uint8_t* m = mask + (long)s * Cc * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[s][j] = m[j] ? acc[s][j] * keep_scale : 0.f;
    } else if (v.gen == 2) {   // jax.random.bernoulli(key of this camera, keep, (rows, Cc * 8)): element (row, c * 8 + j)
      const uint32_t k0 = v.tf_key[blockIdx.y][0], k1 = v.tf_key[blockIdx.y][1];
      const uint64_t rowlen = (uint64_t)Cc * 8ull, total = (uint64_t)v.tf_rows * rowlen;
      const uint64_t e0 = (uint64_t)(v.tf_row0 + n0 + s) * rowlen + (uint64_t)c * 8ull;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[s][j] = bits_to_unit(random_bits_at(k0, k1, total, e0 + (uint64_t)j)) < keep_p ? acc[s][j] * keep_scale : 0.f;
    } else if (v.gen) {
      const uint64_t ctr = (((uint64_t)blockIdx.y * (uint64_t)v.rows_global + (uint64_t)(v.row_offset + n0 + s)) * (uint64_t)Cc + (uint64_t)c) * 3ull;
      const uint64_t h0 = mix64(v.seed ^ mix64(ctr)), h1 = mix64(v.seed ^ mix64(ctr + 1)), h2 = mix64(v.seed ^ mix64(ctr + 2));
      const unsigned u[8] = {(unsigned)(h0 & 0xFFFFFF), (unsigned)((h0 >> 24) & 0xFFFFFF), (unsigned)((h0 >> 48) | ((h1 & 0xFF) << 16)),
                             (unsigned)((h1 >> 8) & 0xFFFFFF), (unsigned)((h1 >> 32) & 0xFFFFFF), (unsigned)((h1 >> 56) | ((h2 & 0xFFFF) << 8)),
                             (unsigned)((h2 >> 16) & 0xFFFFFF), (unsigned)(h2 >> 40)};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[s][j] = u[j] < keep_thr ? acc[s][j] * keep_scale : 0.f;
    }
    *reinterpret_cast<float4*>(f + (long)s * Cc * 8) = make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
    *reinterpret_cast<float4*>(f + (long)s * Cc * 8 + 4) = make_float4(acc[s][4], acc[s][5], acc[s][6], acc[s][7]);
  }
}

int sle_proprio_fwd_multi(const SleFwdArgs* vs, const ProprioArgs* ps, int n, float keep, int N, int HW, int Cc, int groups, long x_gs,
                          long k_gs, long mask_gs, long f_gs, int state_dim, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad instance count %d", n);
  Multi<SleFwdArgs> mv{};
  Multi<ProprioArgs> pv{};
  for (int i = 0; i < n; ++i) {
    mv.v[i] = vs[i];
    if (ps) { pv.v[i] = ps[i]; SERL_REQUIRE(!ps[i].copy_dst || ps[i].copy_cols <= 64, "copy_cols > 64"); }
  }
  const int nb = N <= 64 ? 2 : 8;
  const int nb_sle = cdiv(Cc, 256) * cdiv(N, nb), nb_prop = ps ? cdiv(N, 4) : 0;
  if (nb == 2)
    SERL_LAUNCH_CHAIN(sle_proprio_fwd_kernel<2>, dim3(nb_sle + nb_prop, groups, n), dim3(256), 0, stream, mv, pv, ps ? 1 : 0,
                      1.0f / keep, (unsigned)(keep * 16777216.0f), keep, N, HW, Cc, x_gs, k_gs, mask_gs, f_gs, state_dim, nb_sle);
  else
    SERL_LAUNCH_CHAIN(sle_proprio_fwd_kernel<8>, dim3(nb_sle + nb_prop, groups, n), dim3(256), 0, stream, mv, pv, ps ? 1 : 0,
                      1.0f / keep, (unsigned)(keep * 16777216.0f), keep, N, HW, Cc, x_gs, k_gs, mask_gs, f_gs, state_dim, nb_sle);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// dK partial[split][hw][c][j] = sum_{n in split} x[n][hw][c] * df[n][c*8+j]
__global__ __launch_bounds__(256) void sle_bwd_kernel(const float* x, const float* df, float* partial, int N,
                                                     int HW, int Cc, int nsplit, long x_gs, long df_gs, long part_gs) {
  const int c = blockIdx.x * 256 + threadIdx.x, hw = blockIdx.y, sp = blockIdx.z % nsplit, grp = blockIdx.z / nsplit;
  if (c >= Cc) return;
  x += grp * x_gs; df += grp * df_gs; partial += grp * part_gs;  // grp = camera
  const int per = (N + nsplit - 1) / nsplit;
  const int nb = sp * per, ne = min(N, nb + per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int n = nb; n < ne; ++n) {
    const float xv = x[((long)n * HW + hw) * Cc + c];
    const float4 d0 = *reinterpret_cast<const float4*>(df + (long)n * Cc * 8 + (long)c * 8);
    const float4 d1 = *reinterpret_cast<const float4*>(df + (long)n * Cc * 8 + (long)c * 8 + 4);
    acc[0] += xv * d0.x; acc[1] += xv * d0.y; acc[2] += xv * d0.z; acc[3] += xv * d0.w;
    acc[4] += xv * d1.x; acc[5] += xv * d1.y; acc[6] += xv * d1.z; acc[7] += xv * d1.w;
  }
  float* o = partial + (((long)sp * HW + hw) * Cc + c) * 8;
  *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// the same partial sums, then the sum over the batch splits by the LAST workgroup to arrive at its (camera, pixel, channel
// block) -- see "last-arriver epilogues" above: sc1 partial stores, drain, ticket; sc1 loads in split order on the reducer
__global__ __launch_bounds__(256) void sle_bwd_fused_kernel(const float* x, const float* df, float* partial, int N, int HW, int Cc,
                                                           int nsplit, long x_gs, long df_gs, long part_gs, float* out, long out_gs,
                                                           int* ctr) {
  __shared__ int s_last;
  const int c = blockIdx.x * 256 + threadIdx.x, hw = blockIdx.y, sp = blockIdx.z % nsplit, grp = blockIdx.z / nsplit;
  x += grp * x_gs; df += grp * df_gs; partial += grp * part_gs;  // grp = camera
  const int per = (N + nsplit - 1) / nsplit;
  const int nb = sp * per, ne = min(N, nb + per);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < Cc) {
    for (int n = nb; n < ne; ++n) {
      const float xv = x[((long)n * HW + hw) * Cc + c];
      const float4 d0 = *reinterpret_cast<const float4*>(df + (long)n * Cc * 8 + (long)c * 8);
      const float4 d1 = *reinterpret_cast<const float4*>(df + (long)n * Cc * 8 + (long)c * 8 + 4);
      acc[0] += xv * d0.x; acc[1] += xv * d0.y; acc[2] += xv * d0.z; acc[3] += xv * d0.w;
      acc[4] += xv * d1.x; acc[5] += xv * d1.y; acc[6] += xv * d1.z; acc[7] += xv * d1.w;
    }
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(partial);
    const long o = (((long)sp * HW + hw) * Cc + c) * 8;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4){acc[0], acc[1], acc[2], acc[3]}), rs, (int)(o * 4), 0, kSc1);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, (f32x4){acc[4], acc[5], acc[6], acc[7]}), rs, (int)(o * 4 + 16), 0, kSc1);
  }
  if (!arrive_is_last(ctr + ((long)grp * HW + hw) * gridDim.x + blockIdx.x, nsplit, &s_last)) return;
  if (c >= Cc) return;
  const __amdgpu_buffer_rsrc_t rs = rsrc_of(partial);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nsplit; ++s) {
    const long o = (((long)s * HW + hw) * Cc + c) * 8;
    s0 += ld_sc1(rs, o);
    s1 += ld_sc1(rs, o + 4);
  }
  float* o = out + grp * out_gs + ((long)hw * Cc + c) * 8;
  *reinterpret_cast<f32x4*>(o) = s0;
  *reinterpret_cast<f32x4*>(o + 4) = s1;
}

int sle_bwd_fused(const float* x, const float* df, float* partial, int N, int HW, int Cc, int nsplit, int groups,
                  long x_gs, long df_gs, long part_gs, float* out, long out_gs, int* ctr, hipStream_t stream) {
  SERL_REQUIRE((long)nsplit * HW * Cc * 8 * 4 < (1L << 31), "SLE partial sums exceed 2 GB");
  SERL_LAUNCH_CHAIN(sle_bwd_fused_kernel, dim3(cdiv(Cc, 256), HW, nsplit * groups), dim3(256), 0, stream, x, df, partial, N,
                     HW, Cc, nsplit, x_gs, df_gs, part_gs, out, out_gs, ctr);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int sle_bwd(const float* x, const float* df, float* partial, int N, int HW, int Cc, int nsplit, int groups,
            long x_gs, long df_gs, long part_gs, hipStream_t stream) {
  SERL_LAUNCH_CHAIN(sle_bwd_kernel, dim3(cdiv(Cc, 256), HW, nsplit * groups), dim3(256), 0, stream, x, df, partial, N,
                     HW, Cc, nsplit, x_gs, df_gs, part_gs);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// =============================================================================================
// REDQ target + critic loss (sac.py:142-191).  Single workgroup: deterministic reductions.
//   y[b] = r[b] + discount*mask[b]*min(Qt[i0][b], Qt[i1][b]);  dQ = 2(Q-y)/(E*Bg)
//   scalars[0..2] = sum (Q-y)^2, sum Q, sum y   (local sums; the host / all-reduce normalises)
//   dhead_b = sum dQ  (gradient of the shared head bias)
// =============================================================================================
__global__ __launch_bounds__(256) void critic_loss_kernel(LossArgs L) {
  __shared__ float red[4][256];
  critic_loss_body(L, red);
}

int critic_loss(const float* qt, const float* q, const float* reward, const float* mask, RedqSel sel, int E,
                int B, float discount, float inv_norm, float* y_out, float* dq, float* scalars, float* dbias,
                hipStream_t stream, bool per_member_bias, const float* logp_next, const float* alpha) {
  SERL_REQUIRE(!per_member_bias || E <= 16, "per-member head bias supports ensembles of at most 16 (got %d)", E);
  SERL_REQUIRE(sel.n >= 0 && sel.n <= 16, "critic_subsample_size %d not in [0, 16]", sel.n);
  const LossArgs L{1, qt, q, reward, mask, sel, E, B, discount, inv_norm, y_out, dq, scalars, dbias, per_member_bias ? 1 : 0, logp_next, alpha};
  SERL_LAUNCH_CHAIN(critic_loss_kernel, dim3(1), dim3(256), 0, stream, L);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// =============================================================================================
// tanh-Gaussian policy head (actor_critic_nets.py:179-272, distrax MultivariateNormalDiag + Tanh)
//   std = clip(exp(log_std), std_min, std_max); u = mean + std*eps; a = tanh(u)
//   logp = sum_j(-eps^2/2 - log std - log(2pi)/2) - sum_j 2(log2 - u - softplus(-2u))
// pre: [2][B][A] (mean slab, log_std slab; biases already added)
// =============================================================================================
__global__ void policy_dist_fwd_kernel(Multi<PolicyDistArgs> mv, int B, int A, float std_min, float std_max) {
  const PolicyDistArgs& v = mv.v[blockIdx.x];  // one workgroup per instance
  __shared__ float red[256];
  float local = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
      float mean = v.bias_mean[j], ls = v.bias_ls[j];
      for (int sp = 0; sp < v.S; ++sp) {  // K-split slabs of the head GEMM: [mean | log_std][split][B][A]
        mean += v.slabs[((long)sp * B + b) * A + j];
        ls += v.slabs[(((long)v.S + sp) * B + b) * A + j];
      }
      const float e = v.eps[(long)b * A + j];
      v.pre[(long)b * A + j] = mean;
      v.pre[((long)B + b) * A + j] = ls;
      const float sd = fminf(fmaxf(expf(ls), std_min), std_max);
      const float u = mean + sd * e;
      v.act[(long)b * v.ld_act + j] = tanhf(u);
      v.std_out[(long)b * A + j] = sd;
      lp += -0.5f * e * e - logf(sd) - 0.91893853320467274f;
      lp -= 2.f * (0.69314718055994531f - u - softplusf(-2.f * u));
    }
    v.logp[b] = lp;
    local += lp;
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && v.sum_logp) *v.sum_logp = red[0];
  if (threadIdx.x == 0 && v.alpha_out) v.alpha_out[0] = softplusf(v.lam[0]);
}

int policy_dist_fwd_multi(const PolicyDistArgs* vs, int n, int B, int A, float std_min, float std_max, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad instance count %d", n);
  Multi<PolicyDistArgs> mv{};
  for (int i = 0; i < n; ++i) mv.v[i] = vs[i];
  SERL_LAUNCH_CHAIN(policy_dist_fwd_kernel, dim3(n), dim3(256), 0, stream, mv, B, A, std_min, std_max);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}


// proprio branch (encoding.py:55-70): y = tanh(LayerNorm_1e-6(state W + b)), W [S][64]; one wave per row,
// lane = output feature.  Replaces a GEMM + LN launch pair for this tiny layer.
__global__ __launch_bounds__(256) void proprio_fwd_kernel(Multi<ProprioArgs> mv, int S, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < rows) proprio_row(mv.v[blockIdx.y], S, row, threadIdx.x & 63);
}

int proprio_fwd_multi(const ProprioArgs* vs, int n, int S, int rows, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad instance count %d", n);
  Multi<ProprioArgs> mv{};
  for (int i = 0; i < n; ++i) {
    mv.v[i] = vs[i];
    SERL_REQUIRE(!vs[i].copy_dst || vs[i].copy_cols <= 64, "copy_cols > 64");
  }
  SERL_LAUNCH_CHAIN(proprio_fwd_kernel, dim3(cdiv(rows, 4), n), dim3(256), 0, stream, mv, S, rows);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}


// backward of the actor loss through the distribution:
//   G_u = dL/da*(1-a^2) + c_lp*2a   (d logp/du = 2 tanh u);  dmean = G_u;
//   dstd = G_u*eps - c_lp/std;  dlog_std = dstd*std if std_min < exp(ls) < std_max else 0
// dpre: [2][B][A].  c_lp = dL/dlogp (device scalar *alpha times coef).
__global__ void policy_dist_bwd_kernel(const float* da, long ld_da, const float* act, long ld_act,
                                       const float* pre, const float* stdv, const float* eps,
                                       const float* alpha, float coef, int B, int A, float std_min,
                                       float std_max, float* dpre, const float* q, int E, float* qmean_out) {
  if (blockIdx.x == gridDim.x - 1) {  // rider (one extra workgroup): sum_b mean_e Q[e][b], the actor-loss info scalar
    __shared__ float red[256];
    float sacc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
      float m = 0.f;
      for (int e = 0; e < E; ++e) m += q[(long)e * B + b];
      sacc += m / (float)E;
    }
    red[threadIdx.x] = sacc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) qmean_out[0] = red[0];
    return;
  }
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * A) return;
  const int b = e / A, j = e - b * A;
  const float c_lp = alpha[0] * coef;
  const float a = act[(long)b * ld_act + j];
  const float gu = da[(long)b * ld_da + j] * (1.f - a * a) + c_lp * 2.f * a;
  const float sd = stdv[e];
  const float raw = expf(pre[((long)B + b) * A + j]);
  const float dstd = gu * eps[e] - c_lp / sd;
  dpre[e] = gu;
  dpre[(long)B * A + e] = (raw > std_min && raw < std_max) ? dstd * sd : 0.f;
}

int policy_dist_bwd(const float* da, long ld_da, const float* act, long ld_act, const float* pre,
                    const float* stdv, const float* eps, const float* alpha, float coef, int B, int A,
                    float std_min, float std_max, float* dpre, const float* q, int E, float* qmean_out,
                    hipStream_t stream) {
  SERL_LAUNCH_CHAIN(policy_dist_bwd_kernel, dim3(cdiv(B * A, 256) + 1), dim3(256), 0, stream, da, ld_da, act,
                     ld_act, pre, stdv, eps, alpha, coef, B, A, std_min, std_max, dpre, q, E, qmean_out);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// small helpers ---------------------------------------------------------------------------------
__global__ void copy_cols_multi_kernel(Multi<CopyJob> mv, int rows) {
  const CopyJob& j = mv.v[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * j.cols) return;
  const int r = e / j.cols, c = e - r * j.cols;
  j.dst[(long)r * j.ld_dst + c] = j.src[(long)r * j.ld_src + c];
}
int copy_cols_multi(const CopyJob* jobs, int n, int rows, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad job count %d", n);
  Multi<CopyJob> mv{};
  int cmax = 0;
  for (int i = 0; i < n; ++i) { mv.v[i] = jobs[i]; cmax = std::max(cmax, jobs[i].cols); }
  SERL_LAUNCH_CHAIN(copy_cols_multi_kernel, dim3(cdiv((long)rows * cmax, 256), n), dim3(256), 0, stream, mv, rows);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

__global__ void fill_kernel(float* p, float v, long n) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) p[e] = v;
}
int fill(float* p, float v, long n, hipStream_t stream) {
  SERL_LAUNCH_CHAIN(fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, p, v, n);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// =============================================================================================
// 3x Adam over the full tree (restricted to each optimizer's non-zero support) + summed update +
// target EMA.  common.py:124-168, optimizers.py:32-46, optax.adam(b1 .9, b2 .999, eps 1e-8).
// A tx that is not in networks_to_update still steps with g = 0 (sac.py:276-277): its moments
// decay and its momentum keeps moving the parameters.
// =============================================================================================
// one element: the frozen leaves past P, the temperature at P - 1, everything a 4-wide vector cannot take
__device__ __forceinline__ void adam_elem(const AdamArgs& a, long i, float cs_c, float cs_a) {
  // adamw (optimizers.py:39-42): every optimizer with a weight decay adds -lr*wd*p on EVERY leaf of the tree
  const float wd_total = a.lr_c * a.wd_c + a.lr_a * a.wd_a + a.lr_t * a.wd_t;
  if (i >= a.P) {  // frozen-trunk leaves: no gradient ever reaches them; weight decay and the target EMA still do
    const long k = i - a.P;
    if (k < a.n_frozen) {
      float p = a.frozen[k];
      if (wd_total != 0.f) { p -= wd_total * p; a.frozen[k] = p; }
      if (a.ema_on) a.frozen_target[k] = p * a.tau + a.frozen_target[k] * (1.f - a.tau);
    }
    return;
  }
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  const float p0 = a.theta[i];
  float ua = 0.f, uc = 0.f, ut = 0.f;
  if (i < a.Pc) {
    const float g = a.critic_on ? a.g_critic[i] * cs_c : 0.f;
    const float m = b1 * a.m_c[i] + (1.f - b1) * g;
    const float v = b2 * a.v_c[i] + (1.f - b2) * g * g;
    a.m_c[i] = m; a.v_c[i] = v;
    uc = -a.lr_c * (m / a.bc1) / (sqrtf(v / a.bc2) + eps);
  }
  if (i >= a.Pa0 && i < a.Pa1) {
    const long k = i - a.Pa0;
    const float g = a.actor_on ? a.g_actor[k] * cs_a : 0.f;
    const float m = b1 * a.m_a[k] + (1.f - b1) * g;
    const float v = b2 * a.v_a[k] + (1.f - b2) * g * g;
    a.m_a[k] = m; a.v_a[k] = v;
    ua = -a.lr_a * (m / a.bc1) / (sqrtf(v / a.bc2) + eps);
  }
  if (i == a.P - 1) {  // temperature multiplier (lagrange.py): dL/dlambda = sigmoid(lambda)*(H - H_target)
    float g = 0.f;
    if (a.temp_on) {
      const float H = -a.sum_logp_next[0] * a.inv_batch;
      g = (1.f / (1.f + expf(-p0))) * (H - a.target_entropy);
      a.temp_grad_out[0] = g;
      if (a.clip_t > 0.f && !(fabsf(g) < a.clip_t)) g = g * a.clip_t / fabsf(g);
    }
    const float m = b1 * a.m_t[0] + (1.f - b1) * g;
    const float v = b2 * a.v_t[0] + (1.f - b2) * g * g;
    a.m_t[0] = m; a.v_t[0] = v;
    ut = -a.lr_t * (m / a.bc1) / (sqrtf(v / a.bc2) + eps);
  }
  // optax.adamw: -lr * (adam direction + wd * p); the three optimizers' updates are summed (common.py:161-164)
  if (a.wd_a != 0.f) ua -= a.lr_a * a.wd_a * p0;
  if (a.wd_c != 0.f) uc -= a.lr_c * a.wd_c * p0;
  if (a.wd_t != 0.f) ut -= a.lr_t * a.wd_t * p0;
  const float p = p0 + ((ua + uc) + ut);
  a.theta[i] = p;
  if (a.ema_on) a.theta_target[i] = p * a.tau + a.theta_target[i] * (1.f - a.tau);
}

// A thread owns FOUR consecutive leaves.  A vector that lies wholly inside or outside each optimizer's support (and does not
// hold the temperature) takes the fast path: its eight 16-byte loads are issued together, unconditionally (a vector outside a
// support reads the support's first elements and ignores them), then the arithmetic of adam_elem per element, then the stores.
// The one-leaf-per-thread kernel of rounds 1-3 put every 4-byte load into its own exec-masked region behind stores that were
// still in flight (one vmcnt for both): four dependent round trips per thread, 45 us for the critic step.
__global__ __launch_bounds__(256) void adam_ema_kernel(AdamArgs a) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t == 0 && a.info_mode) {  // info dict of this step (riding along: one launch less on the chain)
    const float* sc = a.scalars;
    float* acc = a.info_acc;
    if (a.info_mode & 1) {  // critic step (sac.py:118-191); weighted mean over UTD minibatches
      if (a.info_reset)
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
      acc[0] += a.info_w * sc[0] * a.inv_eb;   // critic_loss
      acc[1] += a.info_w * sc[1] * a.inv_eb;   // predicted_qs
      acc[2] += a.info_w * sc[2] * a.inv_batch;  // target_qs
    }
    if (a.info_mode & 2) {  // actor + temperature step (sac.py:193-234)
      const float alpha = a.alpha[0];
      acc[3] = -(sc[3] - alpha * sc[4]) * a.inv_batch;  // actor_loss
      acc[4] = alpha;                                   // temperature
      acc[5] = -sc[4] * a.inv_batch;                    // entropy
      acc[6] = alpha * (-sc[5] * a.inv_batch - a.target_entropy);  // temperature_loss
    }
  }
  // clip_by_global_norm (optimizers.py:36-37): g <- g * max_norm / ||g|| when ||g|| >= max_norm
  float cs_c = 1.f, cs_a = 1.f;
  if (a.clip_c > 0.f && a.critic_on) { const float n = sqrtf(a.norm2[0]); if (!(n < a.clip_c)) cs_c = a.clip_c / n; }
  if (a.clip_a > 0.f && a.actor_on) { const float n = sqrtf(a.norm2[1]); if (!(n < a.clip_a)) cs_a = a.clip_a / n; }
  const long i4 = 4 * t;
  const bool c_all = i4 + 3 < a.Pc, c_none = i4 >= a.Pc;
  const bool a_all = i4 >= a.Pa0 && i4 + 3 < a.Pa1, a_none = i4 + 3 < a.Pa0 || i4 >= a.Pa1;
  if (a.vec_ok && i4 + 3 < a.P - 1 && (c_all || c_none) && (a_all || a_none)) {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const long ic = c_all ? i4 : 0, ka = a_all ? i4 - a.Pa0 : 0;
    const float4 p0 = *reinterpret_cast<const float4*>(a.theta + i4);
    const float4 tt = *reinterpret_cast<const float4*>((a.ema_on ? a.theta_target : a.theta) + i4);
    const float4 gc = *reinterpret_cast<const float4*>(a.g_critic + ic);
    const float4 mc = *reinterpret_cast<const float4*>(a.m_c + ic), vc = *reinterpret_cast<const float4*>(a.v_c + ic);
    // (the actor support starts at an arbitrary leaf offset -- behind the critic head's 1-element bias: its arrays are indexed by
    //  i - Pa0 and only 4-byte aligned; global 16-byte accesses need no more)
    const f32x4 ga = *reinterpret_cast<const f32x4u*>(a.g_actor + ka);
    const f32x4 ma = *reinterpret_cast<const f32x4u*>(a.m_a + ka), va = *reinterpret_cast<const f32x4u*>(a.v_a + ka);
    const float p0e[4] = {p0.x, p0.y, p0.z, p0.w}, tte[4] = {tt.x, tt.y, tt.z, tt.w};
    const float gce[4] = {gc.x, gc.y, gc.z, gc.w}, mce[4] = {mc.x, mc.y, mc.z, mc.w}, vce[4] = {vc.x, vc.y, vc.z, vc.w};
    const float gae[4] = {ga[0], ga[1], ga[2], ga[3]}, mae[4] = {ma[0], ma[1], ma[2], ma[3]}, vae[4] = {va[0], va[1], va[2], va[3]};
    float pn[4], tn[4], mcn[4], vcn[4], man[4], van[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ua = 0.f, uc = 0.f;
      {
        const float g = a.critic_on ? gce[e] * cs_c : 0.f;
        const float m = b1 * mce[e] + (1.f - b1) * g;
        const float v = b2 * vce[e] + (1.f - b2) * g * g;
        mcn[e] = m; vcn[e] = v;
        if (c_all) uc = -a.lr_c * (m / a.bc1) / (sqrtf(v / a.bc2) + eps);
      }
      {
        const float g = a.actor_on ? gae[e] * cs_a : 0.f;
        const float m = b1 * mae[e] + (1.f - b1) * g;
        const float v = b2 * vae[e] + (1.f - b2) * g * g;
        man[e] = m; van[e] = v;
        if (a_all) ua = -a.lr_a * (m / a.bc1) / (sqrtf(v / a.bc2) + eps);
      }
      float ut = 0.f;
      if (a.wd_a != 0.f) ua -= a.lr_a * a.wd_a * p0e[e];
      if (a.wd_c != 0.f) uc -= a.lr_c * a.wd_c * p0e[e];
      if (a.wd_t != 0.f) ut -= a.lr_t * a.wd_t * p0e[e];
      pn[e] = p0e[e] + ((ua + uc) + ut);
      tn[e] = pn[e] * a.tau + tte[e] * (1.f - a.tau);
    }
    if (c_all) {
      *reinterpret_cast<float4*>(a.m_c + i4) = make_float4(mcn[0], mcn[1], mcn[2], mcn[3]);
      *reinterpret_cast<float4*>(a.v_c + i4) = make_float4(vcn[0], vcn[1], vcn[2], vcn[3]);
    }
    if (a_all) {
      *reinterpret_cast<f32x4u*>(a.m_a + ka) = (f32x4){man[0], man[1], man[2], man[3]};
      *reinterpret_cast<f32x4u*>(a.v_a + ka) = (f32x4){van[0], van[1], van[2], van[3]};
    }
    *reinterpret_cast<float4*>(a.theta + i4) = make_float4(pn[0], pn[1], pn[2], pn[3]);
    if (a.ema_on) *reinterpret_cast<float4*>(a.theta_target + i4) = make_float4(tn[0], tn[1], tn[2], tn[3]);
    return;
  }
  const long total = a.P + a.n_frozen_live;
#pragma unroll 1
  for (int e = 0; e < 4; ++e)
    if (i4 + e < total) adam_elem(a, i4 + e, cs_c, cs_a);
}

// `steps` deferred target-EMA steps of the frozen trunk leaves at once (common.py:124-134 on leaves no gradient or weight decay
// ever touches): t <- p*tau + t*(1-tau), repeated -- the same expression, hence the same rounding sequence, as adam_ema_kernel
// applies step by step.  The iteration stops at its fixed point (t no longer changes), which a constant p reaches after a few steps.
__global__ __launch_bounds__(256) void frozen_ema_kernel(const float* frozen, float* frozen_target, long n, float tau, long steps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p = frozen[i];
  float t = frozen_target[i];
  for (long k = 0; k < steps; ++k) {
    const float u = p * tau + t * (1.f - tau);
    if (u == t) break;
    t = u;
  }
  frozen_target[i] = t;
}
int frozen_ema(const float* frozen, float* frozen_target, long n, float tau, long steps, hipStream_t stream) {
  if (n <= 0 || steps <= 0) return SERL_OK;
  SERL_LAUNCH_CHAIN(frozen_ema_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, frozen, frozen_target, n, tau, steps);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int adam_ema(const AdamArgs& a, hipStream_t stream) {
  ProfScope prof("adam_ema", stream);
  const bool frozen = a.n_frozen > 0 && (a.ema_on || a.wd_c != 0.f || a.wd_a != 0.f || a.wd_t != 0.f);
  AdamArgs v = a;
  v.n_frozen_live = frozen ? a.n_frozen : 0;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  v.vec_ok = a.Pc >= 4 && a.Pa1 - a.Pa0 >= 4 && al16(a.theta) && al16(a.theta_target) && al16(a.g_critic) && al16(a.m_c) &&
             al16(a.v_c) && a.g_critic && a.g_actor && a.m_a && a.v_a;
  SERL_LAUNCH_CHAIN(adam_ema_kernel, dim3(cdiv(cdiv(a.P + v.n_frozen_live, 4), 256)), dim3(256), 0, stream, v);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// squared global norms of the two gradient ranges (clip_by_global_norm): one 1024-thread block per range, fp64 partials
__global__ __launch_bounds__(1024) void grad_norm2_kernel(const float* gc, long nc, const float* ga, long na, float* out) {
  const float* g = blockIdx.x ? ga : gc;
  const long n = blockIdx.x ? na : nc;
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) s += (double)g[i] * (double)g[i];
  __shared__ double red[1024];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)red[0];
}
int grad_norm2(const float* g_critic, long nc, const float* g_actor, long na, float* out, hipStream_t stream) {
  SERL_LAUNCH_CHAIN(grad_norm2_kernel, dim3(2), dim3(1024), 0, stream, g_critic, nc, g_actor, na, out);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

// =============================================================================================
// device noise (production mode): counter-based hash -> N(0,1) and Bernoulli(keep) masks
// =============================================================================================
__device__ __forceinline__ void gen_one(const NoiseJob& j, long i0) {
  if (i0 >= j.n) return;
  long i = i0;  // position in the global tensor
  if (j.rows_global) {
    const long e = i0 % j.row_elems, rr = i0 / j.row_elems, r = rr % j.rows_local, plane = rr / j.rows_local;
    i = (plane * j.rows_global + j.row_offset + r) * j.row_elems + e;
  }
  if (j.kind == 0) {
    const uint64_t r = mix64(j.seed ^ mix64((uint64_t)i));
    const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(uint32_t)((r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    static_cast<float*>(j.out)[i0] = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
  } else {
    const uint64_t r = mix64(j.seed ^ mix64((uint64_t)i + 0x51ED270Bull));
    static_cast<uint8_t*>(j.out)[i0] = ((float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f)) < j.keep ? 1 : 0;
  }
}
// all the noise of one update phase in one launch: blockIdx.y = job (a normal tensor or a keep-mask)
__global__ void gen_noise_kernel(Multi<NoiseJob> mv) {
  gen_one(mv.v[blockIdx.y], (long)blockIdx.x * 256 + threadIdx.x);
}
int gen_noise_multi(const NoiseJob* vs, int n, hipStream_t stream) {
  SERL_REQUIRE(n >= 1 && n <= kMaxMulti, "bad job count %d", n);
  Multi<NoiseJob> mv{};
  long nmax = 0;
  for (int i = 0; i < n; ++i) { mv.v[i] = vs[i]; nmax = std::max(nmax, vs[i].n); }
  SERL_LAUNCH_CHAIN(gen_noise_kernel, dim3(cdiv(nmax, 256), n), dim3(256), 0, stream, mv);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}
}  // namespace serl
