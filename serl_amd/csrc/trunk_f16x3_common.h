// Part of the split-fp16 trunk (trunk_f16x3.hip includes these in order; round 6 split the 2,600-line file by kernel family):
// shared by every split-fp16 trunk kernel: argument structs, the hi / lo' split, LDS swizzle, the tile tickets and the
// arrive-and-wait of the fused GroupNorm epilogues, the C-layout fused store (register-staged and LDS-DMA kernels).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "prof.h"
#include "trunk_common.h"

namespace serl {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;  // 2^11

// GroupNorm (+ residual) + ReLU + split8 re-layout in the PRODUCING conv's epilogue (mode != 0) instead of a separate
// elementwise pass over the raw fp32 tensor.  GroupNorm needs the statistics of the whole image, which G = 2..8 workgroups
// produce: each adds its partial sums (fp64 atomics, as before), then bumps an arrival counter of the image and waits
// until all of the image's workgroups have arrived.
// FORWARD PROGRESS.  Tiles are handed out by atomic TICKETS taken when a workgroup starts running (fused_tile): one
// counter per XCD, each covering a contiguous range of whole images, so the G tiles of an image carry CONSECUTIVE tickets
// of one counter (and are fetched through one L2).  A waiting workgroup therefore waits (a) for tiles that running
// workgroups already hold -- they finish without waiting for anybody -- or (b) for not-yet-taken tiles of the ONE image
// per counter that straddles its next ticket; at most G - 1 workgroups per counter can wait in state (b), so as long as
// more than 8 (G - 1) workgroups are resident, one of them is running or about to start and takes the missing tickets
// (a workgroup whose own XCD's range is used up takes from the next XCD's counter).  The launcher checks that bound
// against the CUs the stream may use (resident_workgroups) and falls back to the separate elementwise pass otherwise;
// the spin itself is bounded (trap) so a protocol error aborts the kernel instead of hanging the GPU.
// Wave priority of the trunk's conv kernels (s_setprio 3).  In the pipelined step the frozen trunk's stream IS the critical path and the
// update chain's workgroups share its SIMDs (they are sized to fit beside two trunk workgroups per CU): the arbiter then prefers the
// trunk's waves.  Same-call A/B (profiles/r05_ab_wave_prio.txt): pipelined 2.4176 / 2.4124 -> 2.4004 / 2.3992 ms (stage-0 convs
// -13 .. -22 us, conv_init -20 us; the chain's kernels move under the later convs, +6 .. +10 us there), serial unchanged, one rank of
// eight (128 images per pass, where the CHAIN is the critical path) 0.6788 -> 0.6862: on from 512 images per pass.  Value 2 (the default
// when on): 3 in the main loop, 1 in the block convs' epilogues -- an epilogue (HBM traffic, conversions, the wait for the image's other
// tiles) then yields the SIMD to the main loop of the CU's other workgroup: 2.391 / 2.4062 (flat 3) -> 2.3824 / 2.3987, same call.
// (SERL_TRUNK_WPRIO, the switch that forced off / flat / main-loop-over-epilogue, went with round 6: the policy is fixed.)
static int trunk_wave_prio(long images) { return images >= 512 ? 2 : 0; }

struct FuseArgs {
  int mode;                 // 0 off; 1 relu(GN(y)); 2 relu(GN(y) + res_split); 3 relu(GN(y) + GN_res(res_raw))
  int expected;             // arrivals per counter; 0 = LOCAL: a wave's 64 rows x 64 columns are exactly one (image, group), no
                            // workgroup exchanges anything (P == 64 and Cout / 4 == 64: stage 2) -- no ticket, no wait
  int* sync;                // [image][tiles_n] arrival counters, zeroed with the statistics
  int* ticket;              // [8] per-XCD tile counters, zeroed with the statistics
  int group;                // G: tiles (workgroups) per image -- tickets of one image are consecutive
  GnRef gn;                 // this conv's statistics (being produced), scale, bias
  GnRef res_gn;             // mode 3: the projection's GroupNorm (complete: that conv ran before)
  const uint8_t* res_split; // mode 2: the block input (split8)
  const float* res_raw;     // mode 3: raw projection output
  uint8_t* out_split;       // split8 output
  float* out_f32;           // instead of out_split when set: plain fp32 NHWC output (the last block's conv1 writes the trunk's features)
};

struct ConvArgsB {
  ConvArgs c;           // .w unused
  FuseArgs fz;
  const uint16_t* whi;  // [Cout][K]
  const uint16_t* wlo;
  const float* winv;    // [Cout] 1 / (per-output-channel weight scale)
  const uint16_t* wdma;  // LDS-DMA kernel: the planes in its piece order (pack_dma_order_f16x3) or nullptr
  int K;
  // K-split of the small-M register-staged kernel (a rank's share of a data-parallel batch): `ksplit` workgroups per 64x64
  // tile, each over a contiguous range of K chunks; partial tiles go to `kslab` [tile][split][4 waves][4 quads][64 lanes][4]
  // (the accumulator registers as they are: 16-byte write-through stores) and the workgroup that arrives last at `kctr[tile]`
  // adds them in split order and runs the ordinary epilogue (raw store + statistics)
  int ksplit;
  float* kslab;
  int* kctr;
  // row-slab kernel, fused epilogue: the SECOND workgroup of every CU (block ids 256..511 of the first round) starts
  // `stagger` x s_sleep(127) late (before it takes its tile ticket), see the kernel
  int stagger;
  int wprio;   // wave priority (s_setprio) of the kernel's waves: the frozen trunk is the step's critical path, the chain's waves that share a SIMD are not
};

// LDS-DMA kernel, default since round 5 (SERL_PROJ_FUSE=0 switches it off): the block's 1x1 stride-2 projection computed by the SAME workgroup in front of its
// 3x3 stride-2 conv0 tile (same input, same output tile: the projection's pixel is conv0's tap (0, 0)).  A separate kernel
// parameter behind the existing ones, and a separate instantiation (PROJ): the kernels without it keep their code and their
// argument offsets.
struct ConvProjB {
  const uint16_t* wdma;  // the projection's planes in piece order (K = Cin)
  const float* winv;     // [Cout]
  float* out;            // raw fp32 [M][Cout]
  double* stats;         // [N][4][2]
};

typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float clamp_h(float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }

// float4 -> 4 fp16 hi (packed in uint2) and 4 fp16 lo' = fp16((x - hi) * 2^11).
// hi is converted with v_cvt_pkrtz (any hi within one fp16 ulp works: the residual is exact in fp32 and
// stays in range after the 2^11 scale); lo' is rounded to nearest (v_cvt_pk_f16_f32), so
// |x - hi - 2^-11 lo'| <= 2^-21 |x|.  Activations are GroupNorm outputs (|x| << 65504): no clamp here,
// the one-time weight packing clamps.
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const h16x2 h0 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
  const h16x2 h1 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
  const f32x2 r0 = {(v.x - (float)h0[0]) * kLoScale, (v.y - (float)h0[1]) * kLoScale};
  const f32x2 r1 = {(v.z - (float)h1[0]) * kLoScale, (v.w - (float)h1[1]) * kLoScale};
  const f16x2 l0 = __builtin_convertvector(r0, f16x2), l1 = __builtin_convertvector(r1, f16x2);
  hi = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

// byte offset of 16-byte slot `slot` (0..3) of row `row` in a [rows][32] bf16 plane (64-byte rows)
__device__ __forceinline__ int swz(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// Tile of this workgroup in a fused launch of `ntiles` = gridDim.x tiles, `group` tiles per image.  The images are
// split into 8 contiguous ranges (one per XCD, as xcd_remap does for block ids); a workgroup draws from the counter of the
// XCD it actually runs on (HW_REG_XCC_ID -- used for L2 affinity only, any value 0..7 is correct) and moves on to the next
// XCD's counter when that range is used up.  #workgroups == #tiles and every valid ticket is unique, so every workgroup
// finds a tile within one round over the 8 counters.
__device__ __forceinline__ int fused_tile(const FuseArgs& fz, int ntiles) {
  __shared__ int s_tile;
  if (threadIdx.x == 0) {
    const int G = fz.group, ngroups = ntiles / G, gq = ngroups >> 3, gr = ngroups & 7;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7u;
    int tile = -1;
    for (int k = 0; k < 8 && tile < 0; ++k, x = (x + 1) & 7u) {
      const int g0 = (int)x < gr ? (int)x * (gq + 1) : gr * (gq + 1) + ((int)x - gr) * gq;
      const int cnt = (gq + ((int)x < gr ? 1 : 0)) * G;
      if (cnt == 0) continue;
      const int t = __hip_atomic_fetch_add(fz.ticket + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (t < cnt) tile = g0 * G + t;
    }
    if (tile < 0) __builtin_trap();   // cannot happen: as many workgroups as tiles
    s_tile = tile;
  }
  __syncthreads();
  return s_tile;
}

// Ordering without cache maintenance: the statistics, the arrival counters and the tickets are only ever touched by
// SYSTEM-scope atomics (sc1: performed at the memory side, past the 8 per-XCD L2s -- an image's workgroups can sit on
// different XCDs, and agent-scope atomics performed in one XCD's L2 reached the others late: 1e-4 errors at 1024 images),
// and pollers read the statistics with system-scope atomic loads, so there is no cached copy anywhere that an L2
// write-back / L1 invalidate would have to refresh (an agent-scope ACQUIRE in the polling loop invalidates caches on every
// poll: measured 4x slower convs).  What remains is the ORDER "statistics performed before the arrival is performed":
//   * the statistics atomics are RETURNING atomics whose results are consumed (stats_flush): a wave passes the
//     s_waitcnt in front of the barrier below only when the memory side has answered, i.e. performed, each of them.
//     (A NO-RETURN atomic leaves vmcnt when the L2 has ACCEPTED it -- trunk_common.h -- which is why the round-2
//     no-return variant lost sums at 1024 images.)
//   * the barrier orders every wave's (performed) statistics before thread 0 issues the arrival atomic.
// In HIP memory-model terms the arrival is the release and the poll that sees `expected` the acquire; relaxed atomics
// are enough here because every location involved is accessed with memory-side atomics only -- this rests on the measured
// gfx950 behaviour above (tests/test_agent_gpu.py::test_fused_groupnorm_epilogue_is_race_free_*), not on the language model.
__device__ __forceinline__ void fused_arrive_and_wait(int* ctr, int expected) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (the bound turns a protocol error into a kernel abort instead of a hung GPU; a real wait is a few microseconds)
    for (int spins = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < expected; ++spins) {
      __builtin_amdgcn_s_sleep(4);
      if (spins > (1 << 22)) __builtin_trap();
    }
  }
  __syncthreads();
}

// gn_coef4 for one channel; LIVE: the statistics were written by other workgroups of this launch (read at L2)
template <bool LIVE>
__device__ __forceinline__ void gn_coef1(const GnRef& g, int n, int c, float& sc, float& sh) {
  const double* st = g.stats + ((size_t)n * kGnGroups + c / g.gsize) * 2;
  const double s0 = LIVE ? __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : st[0];
  const double s1 = LIVE ? __hip_atomic_load(st + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : st[1];
  const double mean = s0 * g.inv_count, m2 = s1 * g.inv_count;
  const float var = fmaxf((float)(m2 - mean * mean), 0.f);
  const float rstd = rsqrtf(var + 1e-5f), mf = (float)mean;
  sc = g.gamma[c] * rstd;
  sh = g.beta[c] - mf * sc;
}

__device__ __forceinline__ uint32_t swap_adjacent_lanes(uint32_t v) {   // quad_perm [1, 0, 3, 2]
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}
__device__ __forceinline__ float half_bits_to_float(uint32_t b) {
  return (float)__builtin_bit_cast(_Float16, (uint16_t)(b & 0xffffu));
}

// The MFMA C layout gives a lane ONE channel (col0 + 32 tn + li) of 16 rows per 32x32 tile; the split8 layout wants
// the 8 hi halves of 8 consecutive channels in one 16-byte unit and their lo' halves in the next.  Adjacent lanes
// (channels c, c+1) trade halves: the even lane ends up with the dword of the two hi halves, the odd lane with the dword
// of the two lo' halves, so 32 lanes write the same contiguous 128 bytes a row of 32 fp32 values took.  Two rows are
// processed together (packed fp32 math, one cvt_pkrtz / cvt_pk per pair, ONE lane exchange per pair): the epilogue's
// VALU work competes with the other workgroup's MFMAs on the same SIMD, so instruction count matters here.
template <int TM, int TN>
struct FusedResidual { uint32_t v[TM][TN][16]; };

// residual operand of this lane's elements, loaded BEFORE the statistics wait so the latency hides behind it
template <int TM, int TN>
__device__ __forceinline__ void fused_load_residual(const ConvArgsB& ab, FusedResidual<TM, TN>& res, int wrow0, int col0,
                                                    int li, int lh) {
  const FuseArgs& fz = ab.fz;
  if (fz.mode < 2) return;
  const int Cout = ab.c.Cout;
  const bool odd = li & 1;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wrow0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const size_t rowb = (size_t)m * Cout * 4;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int c = col0 + tn * 32 + li;
        if (fz.mode == 2) res.v[tm][tn][r] = *reinterpret_cast<const uint32_t*>(fz.res_split + rowb + (c & ~7) * 4 + (odd ? 16 : 0) + (c & 6) * 2);
        else res.v[tm][tn][r] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(fz.res_raw) + rowb + c * 4);
      }
    }
}

// NSLOT = 4 (stage 3, images of 16 pixels): the wave's 64 rows are FOUR images, slot s = rows 16 s .. 16 s + 15 = image n_img + s, each with
// its own statistics (local_mean / local_rstd then point at NSLOT values); NSLOT = 1: the whole block lies in image n_img.
template <int TM, int TN, int NSLOT = 1>
__device__ __forceinline__ void fused_gn_store(const ConvArgsB& ab, const f32x16 (&acc)[TM][TN],
                                               const FusedResidual<TM, TN>& res, int n_img, int wrow0, int col0, int li, int lh,
                                               bool local = false, const float* local_mean = nullptr, const float* local_rstd = nullptr) {
  const FuseArgs& fz = ab.fz;
  const int Cout = ab.c.Cout;
  const bool odd = li & 1;
  // v_perm selectors (byte k of the result: 0..3 = bytes of the 2nd operand, 4..7 = bytes of the 1st)
  const uint32_t sel_r0 = odd ? 0x01000504u : 0x05040100u;   // (keep, recv) low halves  -> even: keep|recv<<16, odd: recv|keep<<16
  const uint32_t sel_r1 = odd ? 0x03020706u : 0x07060302u;   // same for the high halves
  const uint32_t sel_lo = 0x05040100u, sel_hi = 0x07060302u; // (a.lo16 | b.lo16 << 16), (a.hi16 | b.hi16 << 16) of perm(b, a, .)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int c = col0 + tn * 32 + li;
    float scs[NSLOT], shs[NSLOT], rss[NSLOT], rhs[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      rss[sl] = 0.f; rhs[sl] = 0.f;
      if (local) {   // statistics of this wave's (or, NSLOT = 4, this workgroup's) own block = the whole (image, group)
        scs[sl] = fz.gn.gamma[c] * local_rstd[sl];
        shs[sl] = fz.gn.beta[c] - local_mean[sl] * scs[sl];
      } else {
        gn_coef1<true>(fz.gn, n_img + sl, c, scs[sl], shs[sl]);
      }
      if (fz.mode >= 3) gn_coef1<false>(fz.res_gn, n_img + sl, c, rss[sl], rhs[sl]);
    }
    const int cbyte = (c & ~7) * 4 + (odd ? 16 : 0) + (c & 6) * 2;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const int r0 = 2 * rp, r1 = r0 + 1;   // rows m and m + 1 (same 16-row slot)
        const int sl = NSLOT == 1 ? 0 : 2 * tm + (r0 >> 3);
        const f32x2 sc2 = {scs[sl], scs[sl]}, sh2 = {shs[sl], shs[sl]}, rs2 = {rss[sl], rss[sl]}, rh2 = {rhs[sl], rhs[sl]};
        const int m = wrow0 + tm * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * lh;
        f32x2 v = (f32x2){acc[tm][tn][r0], acc[tm][tn][r1]} * sc2 + sh2;
        if (fz.mode == 2) {
          const uint32_t o0 = res.v[tm][tn][r0], o1 = res.v[tm][tn][r1];
          // what the neighbour needs from me (even: my hi16 = hi[c+1]; odd: my lo16 = lo[c-1]) and what I keep
          const uint32_t send = __builtin_amdgcn_perm(o1, o0, odd ? sel_lo : sel_hi);
          const uint32_t mine = __builtin_amdgcn_perm(o1, o0, odd ? sel_hi : sel_lo);
          const uint32_t recv = swap_adjacent_lanes(send);
          const uint32_t H = odd ? recv : mine, L = odd ? mine : recv;   // (x_hi row0 | x_hi row1 << 16), same for lo'
          const f16x2 Hh = __builtin_bit_cast(f16x2, H), Lh = __builtin_bit_cast(f16x2, L);
          const f32x2 xh = {(float)Hh[0], (float)Hh[1]}, xl = {(float)Lh[0], (float)Lh[1]};
          v = (xh + xl * (f32x2){kLoInv, kLoInv}) + v;
        } else if (fz.mode == 3) {
          const f32x2 x = {__builtin_bit_cast(float, res.v[tm][tn][r0]), __builtin_bit_cast(float, res.v[tm][tn][r1])};
          v = (x * rs2 + rh2) + v;
        } else if (fz.mode == 4) {   // residual = relu(GroupNorm(raw)): the block input that was never materialised (RAWIN)
          const f32x2 x = {__builtin_bit_cast(float, res.v[tm][tn][r0]), __builtin_bit_cast(float, res.v[tm][tn][r1])};
          const f32x2 y = x * rs2 + rh2;
          v = (f32x2){fmaxf(y[0], 0.f), fmaxf(y[1], 0.f)} + v;
        }
        v = (f32x2){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
        if (fz.out_f32) {   // (uniform) plain fp32 output: the trunk's features
          float* o32 = fz.out_f32 + (size_t)m * Cout + c;
          o32[0] = v[0];
          o32[Cout] = v[1];
          continue;
        }
        const h16x2 hp = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]);
        const f32x2 hf = {(float)hp[0], (float)hp[1]};
        const f16x2 lp = __builtin_convertvector((v - hf) * (f32x2){kLoScale, kLoScale}, f16x2);
        const uint32_t hpb = __builtin_bit_cast(uint32_t, hp), lpb = __builtin_bit_cast(uint32_t, lp);
        const uint32_t keep = odd ? lpb : hpb;
        const uint32_t recv = swap_adjacent_lanes(odd ? hpb : lpb);
        uint8_t* o = fz.out_split + (size_t)m * Cout * 4 + cbyte;
        *reinterpret_cast<uint32_t*>(o) = __builtin_amdgcn_perm(recv, keep, sel_r0);
        *reinterpret_cast<uint32_t*>(o + (size_t)Cout * 4) = __builtin_amdgcn_perm(recv, keep, sel_r1);
      }
  }
}

}  // namespace serl
