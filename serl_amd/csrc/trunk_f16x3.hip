// Split-fp16 ("f16x3") implicit-GEMM conv for the frozen ResNet-10 trunk on MI355X (gfx950).
//
// fp32 operands are split on the fly into  x = hi + 2^-11 * lo'  with hi = fp16(x) and
// lo' = fp16((x - hi) * 2^11)  (the residual is exact in fp32 and the 2^11 scale keeps it in fp16's
// normal range), and every fp32 product is replaced by three fp16 MFMA products accumulated in fp32:
//     a*b ~= a_hi*b_hi + 2^-11 * (a_hi*b_lo' + a_lo'*b_hi)       (the 2^-22 a_lo'*b_lo' term is dropped)
// on v_mfma_f32_32x32x16_f16, whose dense rate is 16x the f32-input MFMA: 3 instructions of 32 cycles
// per 32x32x16 block instead of 8 of 64.  The hi*hi products and the cross products go to separate
// fp32 accumulators that are combined once in the epilogue.  Per-product relative error <= ~3*2^-22
// (7e-7), i.e. fp32-roundoff class; measured error of the whole trunk vs fp64 in tests/test_agent_gpu.py
// next to the exact-fp32 kernel's (every parity test runs in both modes).
//
// Same structure as conv_igemm_kernel (trunk.hip): NHWC activations stay fp32 in HBM, BK = 32 chunks
// inside one (ky,kx) tap, GroupNorm+ReLU of the producer applied on load, GN statistics in the
// epilogue.  Differences: weights are pre-split and pre-transposed once to fp16 [Cout][K] (hi, lo');
// LDS holds fp16 hi/lo' planes with K contiguous (64-byte rows, 16-byte slots XOR-swizzled by
// (row>>2)&3 -> conflict-free ds_read_b128 MFMA fragments).
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
// SERL_TRUNK_WPRIO = 0 / 1 / 2 forces off / flat / main-loop-over-epilogue.
static int trunk_wave_prio(long images) {
  static const int v = []() { const char* e = getenv("SERL_TRUNK_WPRIO"); return e ? atoi(e) : -1; }();
  return v >= 0 ? v : (images >= 512 ? 2 : 0);
}

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
};

struct ConvArgsB {
  ConvArgs c;           // .w unused
  FuseArgs fz;
  const uint16_t* whi;  // [Cout][K]
  const uint16_t* wlo;
  const float* winv;    // [Cout] 1 / (per-output-channel weight scale)
  const uint16_t* wslab; // row-slab kernel: the planes in its fetch order (pack_slab_order_f16x3) or nullptr
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
  // row-slab kernels, fused epilogue: 1 = the tile goes through LDS once and is normalised / stored ROW-major (rowtile_epilogue_t)
  int epi_t;
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

template <int TM, int TN>
__device__ __forceinline__ void fused_gn_store(const ConvArgsB& ab, const f32x16 (&acc)[TM][TN],
                                               const FusedResidual<TM, TN>& res, int n_img, int wrow0, int col0, int li, int lh,
                                               bool local = false, float local_mean = 0.f, float local_rstd = 0.f) {
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
    float sc, sh, rs = 0.f, rh = 0.f;
    if (local) {   // statistics of this wave's own 64 x 64 block = the whole (image, group)
      sc = fz.gn.gamma[c] * local_rstd;
      sh = fz.gn.beta[c] - local_mean * sc;
    } else {
      gn_coef1<true>(fz.gn, n_img, c, sc, sh);
    }
    if (fz.mode >= 3) gn_coef1<false>(fz.res_gn, n_img, c, rs, rh);
    const int cbyte = (c & ~7) * 4 + (odd ? 16 : 0) + (c & 6) * 2;
    const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh}, rs2 = {rs, rs}, rh2 = {rh, rh};
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const int r0 = 2 * rp, r1 = r0 + 1;   // rows m and m + 1
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
        stats_flush(s, q, stp, n0 + wn * WCOLS + tn * 32 + li, gsize, valid);
      }
    }
  }
}

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
  if (PMODE == 0 && ab.fz.mode) fused_load_residual<TM, TN>(ab, fres, wrow0, n0 + wn * WCOLS, li, lh);
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
        stats_flush(s, q, stp, n0 + wn * WCOLS + tn * 32 + li, gsize, valid);
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
    fused_gn_store<TM, TN>(ab, acc, fres, wrow0 / a.P, wrow0, n0 + wn * WCOLS, li, lh, true, (float)mean, rsqrtf(var + 1e-5f));
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
      stats_flush(s, q, stp, n0 + wn * WCOLS + tn * 32 + li, gsize, valid);
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

// Epilogue of a 256 x 64 output tile held by 4 waves of 64 x 64 (all rows in image n_img): combine the two accumulators,
// undo the weight scale, GroupNorm statistics, then either the raw fp32 store or the fused GroupNorm modes.
__device__ __forceinline__ void rowtile_epilogue(const ConvArgsB& ab, f32x16 (&acc)[2][2], f32x16 (&accx)[2][2], int m0, int n0,
                                                 int n_img, int wave, int li, int lh, int sync_idx) {
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
  FusedResidual<TM, TN> fres;
  if (ab.fz.mode) fused_load_residual<TM, TN>(ab, fres, wrow0, n0, li, lh);
  if (!ab.fz.mode) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wrow0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* o = a.out + (size_t)m * a.Cout + n0 + li;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) o[32 * tn] = acc[tm][tn][r];
      }
  }
  {
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
      stats_flush(s, q, stp, n0 + tn * 32 + li, gsize, true);
    }
  }
  if (ab.fz.mode) {
    fused_arrive_and_wait(ab.fz.sync + sync_idx, ab.fz.expected);
    fused_gn_store<TM, TN>(ab, acc, fres, n_img, wrow0, n0, li, lh);
  }
}

// The fused GroupNorm (+ residual) + ReLU + split8 epilogue of a 256 x 64 row tile, ROW-MAJOR (round 5).  rowtile_epilogue above
// stores from the MFMA C layout -- a lane owns ONE channel of 16 rows per 32 x 32 tile, i.e. 64 four-byte stores and (with a
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
    const int gsize = a.Cout / kGnGroups;
    double* stp = a.stats + (size_t)n_img * kGnGroups * 2;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; s += v; q += v * v; }
      stats_flush(s, q, stp, n0 + tn * 32 + li, gsize, true);
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
// RAWIN: the input is the RAW fp32 tensor of the producing layer (conv_init's completed pooling output) and GroupNorm + ReLU +
// the hi / lo' split are applied while a slab is staged (a.in_gn: per-channel scale / shift of this tile's image, held in
// LDS) -- the elementwise pass that would materialise the split8 tensor (read 268 MB + write 268 MB per trunk pass) is gone.
// A thread then stages one (pixel, k-half) = 8 channels per slab part: two 16-byte fp32 loads in, one hi and one lo' unit out.
// WDMA (round 5): the WEIGHTS of a sub-chunk arrive by LDS-DMA in the LDS-DMA kernels' piece order (4 KB per tap, swizzled
// [cout][64 B]: see conv3x3_slabdma_f16x3_kernel) one sub-chunk ahead -- three of a thread's five staging loads, their registers
// and their ds_write_b128 disappear; the activations keep the register path (RAWIN applies GroupNorm + ReLU + split on the way).
template <bool RAWIN, bool WDMA = false, bool EPT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_rowslab_f16x3_kernel(ConvArgsB ab) {
  const ConvArgs& a = ab.c;
  constexpr int TM = 2, TN = 2, WROWS = 64, BM = 256, BN = 64;
  // LDS image: 16-byte units (8 fp16 = one MFMA k-half of one plane) laid out so that the 32 lanes of an MFMA fragment read
  // (consecutive pixels / output channels, same plane and k-half) touch CONSECUTIVE units -- ds_read_b128 serves 16-lane
  // groups over a 256-byte bank row, and [pixel][32 B] rows would put lanes l and l+8 of a group on the same banks (PMC:
  // 45 % of the LDS cycles were bank conflicts with that layout).  Activations: 4 regions (plane, k-half) of [pixel][16 B],
  // region q = plane + 2 * k-half at q * A_REGION; weights per tap: [plane][k-half][cout][16 B].  Region strides are padded
  // so that the 8 lanes of a ds_write_b128 group (2 pixels x 4 regions, or 4 couts x 2 k-halves) cover all 32 write banks.
  constexpr int A_REGION = kRowslabPix * 16 + 32, A_BYTES = 4 * A_REGION;
  constexpr int B_HALF = BN * 16 + 64, B_PLANE = 2 * B_HALF, B_TAP = 2 * B_PLANE, B_BYTES = 3 * B_TAP;
  static_assert(2 * A_BYTES + 2 * B_BYTES == kRowslabLds, "LDS size of the launch");
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
  __shared__ float s_gn[2][128];   // RAWIN: GroupNorm scale / shift per input channel of this tile's image
  if (RAWIN) {
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
  } else {
#pragma unroll
  for (int part = 0; part < 3; ++part)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int u = (part * AJ + j) * 256 + tid, pix = u >> 2, q = u & 3;
      const int sy = pix / pw, sx = pix - sy * pw;
      const int iy = oy0 - 1 + sy, ix = sx - 1;
      const int iyc = min(max(iy, 0), a.Hi - 1), ixc = min(max(ix, 0), a.Wi - 1);
      rbase[part][j] = ((n_img * a.Hi + iyc) * a.Wi + ixc) * a.Cin + 4 * q;
      if (pix < npix && iy == iyc && ix == ixc) okbits |= 1u << (part * AJ + j);
    }
  }
  const int b_plane = tid >> 7, b_cout = (tid >> 1) & 63, b_half = tid & 1;
  const uint16_t* wrow = (b_plane ? ab.wlo : ab.whi) + (size_t)(n0 + b_cout) * ab.K + b_half * 8;
  // fetch-order copy: block (column tile bn, group, tap) of 2048 halfs, this thread's unit at tid * 8
  const uint16_t* wslab = ab.wslab ? ab.wslab + ((size_t)bn * c16n * 9 << 11) + tid * 8 : nullptr;
  u32x4 ra[AJ], rb[3];

// fetch into registers: part PART of the slab of channel group CG (activations), the 3 taps of (channel group BG, row BKY)
#define SERL_RS_LOAD_A(RA, PART, CG)                                                               \
  _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                                   \
    RA[j] = *reinterpret_cast<const u32x4*>(a.in + rbase[PART][j] + ((CG) << 4));
#define SERL_RS_DMA_B(BG, BKY, BBUF)                                                               \
  _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                 \
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(wdsrc + ((size_t)(((BKY) * 3 + kx) * c16n + (BG)) << 12)), \
                                     (lds_void_t*)(smB + (BBUF) * B_BYTES + kx * 4096 + (tid >> 6) * 1024), 16, 0, 0);
#define SERL_RS_LOAD_B(RB, BG, BKY)                                                                \
  if (!WDMA) _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                      \
    RB[kx] = wslab ? *reinterpret_cast<const u32x4*>(wslab + ((size_t)(((BG) * 3 + (BKY)) * 3 + kx) << 11)) \
                   : *reinterpret_cast<const u32x4*>(wrow + ((BKY) * 3 + kx) * a.Cin + ((BG) << 4));
#define SERL_RS_STORE_A(RA, PART, ABUF, CGN)                                                       \
  if (RAWIN) {                                                                                     \
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
  } else                                                                                           \
  _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                 \
    const int u_ = ((PART) * AJ + j) * 256 + tid;                                                  \
    u32x4 v = RA[j];                                                                               \
    if (!((okbits >> ((PART) * AJ + j)) & 1u)) v = (u32x4){0u, 0u, 0u, 0u};                        \
    if (u_ < kRowslabPix * 4)                                                                      \
      *reinterpret_cast<u32x4*>(smA + (ABUF) * A_BYTES + (u_ & 3) * A_REGION + (u_ >> 2) * 16) = v; \
  }
#define SERL_RS_STORE_B(RB, BBUF)                                                                  \
  if (!WDMA) _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                      \
    *reinterpret_cast<u32x4*>(smB + (BBUF) * B_BYTES + kx * B_TAP + b_plane * B_PLANE + b_half * B_HALF + b_cout * 16) = RB[kx];
// what is fetched while sub-chunk (CG, KY) computes: the weights of the NEXT sub-chunk and part KY of the NEXT slab (the
// last slab re-fetches itself: harmless, keeps the loop uniform) -- and where it goes when that sub-chunk is done
#define SERL_RS_LOADS(CG, KY, RA, RB)                                                              \
  {                                                                                                \
    const int ncg_ = (KY) == 2 ? (CG) + 1 : (CG), nky_ = (KY) == 2 ? 0 : (KY) + 1;                 \
    const int ncgc_ = min(ncg_, c16n - 1), sn_ = min((CG) + 1, c16n - 1);                          \
    SERL_RS_LOAD_B(RB, ncgc_, nky_);                                                               \
    if (WDMA) { SERL_RS_DMA_B(ncgc_, nky_, ((CG) * 3 + (KY) + 1) & 1) }                             \
    if ((KY) == 0) { SERL_RS_LOAD_A(RA, 0, sn_); } else if ((KY) == 1) { SERL_RS_LOAD_A(RA, 1, sn_); } else { SERL_RS_LOAD_A(RA, 2, sn_); } \
  }
#define SERL_RS_STORES(C, CG, KY, RA, RB)                                                          \
  {                                                                                                \
    SERL_RS_STORE_B(RB, ((C) + 1) & 1);                                                            \
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
        bhi[tn] = *reinterpret_cast<const f16x8*>(WDMA ? sb + kx * 4096 + wd_bhi[tn] : sb + boff + kx * B_TAP + tn * 32 * 16);          \
        blo[tn] = *reinterpret_cast<const f16x8*>(WDMA ? sb + kx * 4096 + wd_blo[tn] : sb + boff + kx * B_TAP + B_PLANE + tn * 32 * 16); \
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
  const int boff = lh * B_HALF + li * 16;
  int wd_bhi[TN], wd_blo[TN];   // WDMA: swizzled [cout][64 B] image of a tap's weights
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int row = tn * 32 + li, sw = (row >> 2) & 3;
    wd_bhi[tn] = row * 64 + ((lh ^ sw) << 4);
    wd_blo[tn] = row * 64 + (((2 + lh) ^ sw) << 4);
  }
  const uint8_t* wdsrc = reinterpret_cast<const uint8_t*>(ab.wdma) + (size_t)(n0 >> 6) * (9 * c16n) * 4096 + (tid >> 6) * 1024 + lane * 16;
  // prologue: slab 0 (three parts) and the weights of sub-chunk 0
  if (WDMA) { SERL_RS_DMA_B(0, 0, 0) }
  SERL_RS_LOAD_B(rb, 0, 0);
#pragma unroll
  for (int part = 0; part < 3; ++part) {
    SERL_RS_LOAD_A(ra, part, 0);
    SERL_RS_STORE_A(ra, part, 0, 0);
  }
  SERL_RS_STORE_B(rb, 0);
  __syncthreads();
  int cg = 0, ky = 0;   // channel group and kernel row of sub-chunk c
  for (int c = 0; c < nchunks; ++c) {
    SERL_RS_LOADS(cg, ky, ra, rb);
    SERL_RS_COMPUTE(c, cg, ky);
    // the other weight buffer was last read in sub-chunk c - 1, the other slab buffer during the previous channel group
    // (fetching two sub-chunks ahead with a second staging register set was measured neutral in round 2: removed)
    SERL_RS_STORES(c, cg, ky, ra, rb);
    __syncthreads();
    if (++ky == 3) { ky = 0; ++cg; }
  }
#undef SERL_RS_LOAD_A
#undef SERL_RS_LOAD_B
#undef SERL_RS_DMA_B
#undef SERL_RS_STORE_A
#undef SERL_RS_STORE_B
#undef SERL_RS_LOADS
#undef SERL_RS_STORES
#undef SERL_RS_COMPUTE
  if (EPT) rowtile_epilogue_t(ab, acc, accx, m0, n0, n_img, wave, lane, n_img * a.tiles_n + bn, smemb);
  else rowtile_epilogue(ab, acc, accx, m0, n0, n_img, wave, li, lh, n_img * a.tiles_n + bn);
}

// ---------------------------------------------------------------------------------------------
// Row-slab kernel with LDS-DMA staging (round 5; default, SERL_SLAB_DMA=0 = the register-staged kernel): the tile geometry, the K order (16-channel
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
  else rowtile_epilogue(ab, acc, accx, m0, n0, n_img, wave, li, lh, n_img * a.tiles_n + bn);
}

// ---------------------------------------------------------------------------------------------
// conv_init: u8 image -> normalise -> conv 7x7 stride 2 pad 3, 3 -> 64 (+ fused 3x3/2 max-pool).  Persistent workgroups keep
// the weight planes resident in LDS and walk over 16x16 output tiles (conv_init_u8_kernel below).
// ---------------------------------------------------------------------------------------------
struct ConvInitArgsB {
  const uint8_t* img;   // [N][H][W][3]
  const uint16_t* whi;  // [64][224] fp16 (folded, scaled weights: pack_conv_init_u8_kernel)
  const uint16_t* wlo;  // [64][224] fp16 residual (unscaled)
  const float* winv;    // [64] 1 / (per-output-channel weight scale)
  float* out;           // [N][Ho][Wo][64]   (POOL: unused)
  double* stats;        // [N][4][2]
  int N, H, W, Ho, Wo, tiles_y, tiles_x, total_tiles;
  // POOL (fused 3x3/2 max-pool): sign source and the three compact outputs
  const float* gamma;   // [64] GroupNorm scale of norm_init
  float* pooled;        // [N][Ho/2][Wo/2][64] extreme of the in-tile part of every pooling window
  float* first_rows;    // [N][tiles_y][Wo][64] raw conv outputs of rows 0 mod 16
  float* first_cols;    // [N][Ho][tiles_x][64] raw conv outputs of cols 0 mod 16
  int chunk;            // tiles per scheduling chunk (divides tiles_y * tiles_x)
  int* ticket;          // chunk ticket (zeroed per pass)
  int wprio;            // wave priority (s_setprio), see ConvArgsB
  int ablate;           // TIMING EXPERIMENTS ONLY, compiled in with -DSERL_ABLATE (never in the shipped library; SERL_CINIT_ABLATE, results
                        // are wrong): 1 no patch fill, 2 no MFMAs, 4 no pooling epilogue, 8 no pixel fetch
};

constexpr int kCbPatch = 37;     // input rows/cols per 16x16 output tile
// phase ablation of conv_init for timing experiments: a compile-time `false` unless the library is built with -DSERL_ABLATE
__device__ __forceinline__ bool c8_ablate(const ConvInitArgsB& a, int bit) {
#ifdef SERL_ABLATE
  return (a.ablate & bit) != 0;
#else
  (void)a; (void)bit;
  return false;
#endif
}

// POOL: relu(GN(.)) is monotone in the raw conv output with the sign of the channel's GroupNorm scale gamma (a frozen
// parameter), so max_pool(relu(GN(x))) = relu(GN(extreme(x))) with extreme = max where gamma >= 0 and min where
// gamma < 0 -- bit for bit (rounding is monotone).  The pooling can therefore run HERE, before the image's
// statistics exist: the tile writes, per channel, the extreme over the in-tile part of each 3x3/2 window (1/4 of the
// raw tensor) plus its first row and first column raw (the missing row/column of the windows of the tile above /
// to the left), instead of 1 MiB of raw fp32 per image that the pool kernel re-read 1.5x.
// ---------------------------------------------------------------------------------------------
// conv_init on RAW pixels: the ImageNet normalisation is folded into the weights,
//     out = sum_taps_inside ((px/255 - mean_c)/std_c) w  =  sum px * w/(255 std_c)  -  sum_taps_inside (mean_c/std_c) w ,
// so the activation operand is the pixel value itself -- an integer 0..255, EXACT in fp16: it needs no lo' plane and an
// fp32 product costs TWO fp16 MFMA products (px*w_hi + px*w_lo) instead of three.  The second term depends on which
// taps fall inside the image (the reference zero-pads the NORMALISED image, resnet_v1.py:221-223,249-255); it rides in
// the padding lane of the pixel record: a pixel is 4 halfs {c0, c1, c2, 1} (all 0 outside the image) and the weight of
// the 4th lane is -sum_c (mean_c/std_c) w[ky,kx,c,:], so the border-dependent bias comes out of the same MFMAs.
// 8-byte pixels make every 8-wide k-block (two pixels) a 16-byte aligned run of one patch row: K = 7 rows x 8 pixels x 4
// = 224, A fragments are single ds_read_b128 (patch pitch 384 B: the two output rows of a lane group land on
// complementary bank halves -> conflict-free), half the patch bytes of the 3-product kernel.  The folded weights of output
// channel n are scaled by a power of two s_n (largest |w| in [4096, 8192): w_lo stays in fp16's normal range, w_hi cannot
// overflow however strong the filter); the accumulator is rescaled (exactly) by 1/s_n in the epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int kC8K = 224;                    // 7 kernel rows x 8 pixel slots x 4 lanes
constexpr int kC8WP = 232;                   // LDS pitch of a weight row (halfs): 464 B -> conflict-free ds_read_b128
constexpr int kC8Pitch = 384;                // LDS pitch of a patch row (bytes) = 48 pixel slots
constexpr int kC8WBytes = 64 * kC8WP * 2;    // one weight plane
constexpr int kC8PBytes = 14336;             // patch (37 x 384 = 14208 B) / row-exchange buffer of the pooling stage (8 KB); two
                                             // workgroups = 144 KB, which leaves room for one 12 KB update-chain GEMM workgroup
constexpr int kC8Lds = 2 * kC8WBytes + kC8PBytes;
static_assert(kCbPatch * kC8Pitch <= kC8PBytes, "patch does not fit");

// POOL: 0 = raw conv output; 1 = in-tile part of the pooling windows + first rows / columns (completed by
// pool_finish_split_kernel; chunks of 4 tiles: the path for few images); 2 = COMPLETE pooling: a chunk is a whole image walked
// in reverse raster order, so the first row of the tile below and the first column of the tile to the right -- the missing
// third row / column of the windows on this tile's bottom / right edge -- were written by THIS workgroup one to five tiles
// earlier and are read back from L2 (same CU: no cross-XCD coherence involved); no second pass over the pooled tensor.
template <int POOL>
__global__ __launch_bounds__(256, 2) void conv_init_u8_kernel(ConvInitArgsB a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smemb[];
  uint8_t* w_hi = smemb;
  uint8_t* w_lo = smemb + kC8WBytes;
  uint8_t* patch = smemb + 2 * kC8WBytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: `wave == 3` is a uniform branch)
  const int li = lane & 31, lh = lane >> 5;
  if (a.wprio) __builtin_amdgcn_s_setprio(3);   // (3 in the matrix loop only, 1 around it: neutral here, 2.418 / 2.4093 vs 2.4148 / 2.4118)
  for (int v = tid; v < 2 * 64 * (kC8K / 8); v += 256) {   // resident weights: 64 rows x 28 16-byte slots per plane
    const int plane = v / (64 * 28), r = (v / 28) % 64, sl = v % 28;
    const uint4 val = *reinterpret_cast<const uint4*>((plane ? a.wlo : a.whi) + (size_t)r * kC8K + sl * 8);
    *reinterpret_cast<uint4*>((plane ? w_lo : w_hi) + r * (kC8WP * 2) + sl * 16) = val;
  }
  // patch staging: task = (patch row r, group g of 4 image pixels aligned to 4): 12 contiguous image bytes
  constexpr int kGroups = 10, kTasks = kCbPatch * kGroups;   // 370 tasks, 2 rounds of 256 threads
  const long img_bytes = (long)a.N * a.H * a.W * 3;
  const bool aligned = (a.W & 3) == 0 && (reinterpret_cast<uintptr_t>(a.img) & 3) == 0;
  uint32_t pre[2][3];
  unsigned pmask[2];   // bit j: pixel j of the group is inside the image
  const int tpi = a.tiles_y * a.tiles_x;
#define SERL_C8_ORDER(T) (POOL == 2 ? (T) - (T) % tpi + (tpi - 1 - (T) % tpi) : (T))   /* reverse raster inside an image */
#define SERL_C8_FETCH(TILE)                                                                             \
  {                                                                                                     \
    int b_ = SERL_C8_ORDER(TILE);                                                                       \
    const int tx_ = b_ % a.tiles_x;                                                                     \
    b_ /= a.tiles_x;                                                                                    \
    const int ty_ = b_ % a.tiles_y;                                                                     \
    const int n_ = b_ / a.tiles_y;                                                                      \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                     \
      const int t_ = tid + 256 * q;                                                                     \
      const int r_ = t_ / kGroups, g_ = t_ - r_ * kGroups;                                              \
      const int iy = ty_ * 32 - 3 + r_, ixg = tx_ * 32 - 4 + 4 * g_;                                    \
      const bool rowok = t_ < kTasks && (unsigned)iy < (unsigned)a.H;                                   \
      unsigned m_ = 0;                                                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
        if (rowok && (unsigned)(ixg + j) < (unsigned)a.W) m_ |= 1u << j;                                \
      pmask[q] = m_;                                                                                    \
      long off_ = (((long)n_ * a.H + min(max(iy, 0), a.H - 1)) * a.W + ixg) * 3;                        \
      if (aligned) {                                                                                    \
        _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                 \
          const long o_ = min(max(off_ + 4 * d, 0L), img_bytes - 4);                                    \
          pre[q][d] = *reinterpret_cast<const uint32_t*>(a.img + o_);  /* branch-free, see below */     \
        }                                                                                               \
      } else {                                                                                          \
        _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                 \
          uint32_t w_ = 0;                                                                              \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
            const long o_ = min(max(off_ + 4 * d + e, 0L), img_bytes - 1);                              \
            w_ |= (uint32_t)a.img[o_] << (8 * e);                                                       \
          }                                                                                             \
          pre[q][d] = w_;                                                                               \
        }                                                                                               \
      }                                                                                                 \
    }                                                                                                   \
  }
  // (The fetch is BRANCH-FREE: every address is clamped into the image batch and pixels outside the image are zeroed by pmask
  // when the patch is filled.  With `m_ ? load : 0` hipcc put every load into its own exec-masked region and an
  // `s_waitcnt vmcnt(0)` in front of the first one -- which also waits for the previous tile's pooled STORES: the prefetch
  // cost 42 us per pass in a timing ablation.)
  // Tiles are handed out in CHUNKS of a.chunk consecutive tiles of one image (a.chunk divides tiles_per_img): the
  // GroupNorm partial sums stay in registers across a chunk and are flushed once per chunk (per-tile fp64 atomics of 16
  // workgroups on the same 8 words cost 40 us per pass), and neighbouring tiles share their halo in L2.  The first chunk
  // of a workgroup is its block index, the following ones come from an atomic ticket: with a static partition a
  // workgroup that cannot become resident at once (the update chain's kernels own some wave slots when the two streams
  // overlap) starts its whole share late and the kernel takes up to twice as long (measured 259 us alone, 485 us
  // co-running); with tickets a late workgroup simply takes fewer chunks.
  // (An anti-phase start -- the second workgroup of every CU half a tile late -- was worth 3 % of this kernel until the epilogue's
  // stores stopped stalling the next tile's loads; neutral since, removed in round 5.)
  __shared__ int s_next_chunk;
  const int nchunks = a.total_tiles / a.chunk;
  float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
  const float winv[2] = {a.winv[li], a.winv[32 + li]};
  float sgn[2] = {1.f, 1.f};   // sign of the channel's GroupNorm scale (POOL)
  if (POOL) { sgn[0] = a.gamma[li] < 0.f ? -1.f : 1.f; sgn[1] = a.gamma[32 + li] < 0.f ? -1.f : 1.f; }
  int chunk = blockIdx.x, next_chunk = 0;
  int tile = chunk * a.chunk, t_end = tile + a.chunk;
  if (chunk < nchunks) SERL_C8_FETCH(tile);
  while (chunk < nchunks) {
    const bool first_of_chunk = tile == chunk * a.chunk;
    if (first_of_chunk && tid == 0)
      s_next_chunk = (int)gridDim.x + __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int b = SERL_C8_ORDER(tile);
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int oy0 = ty * 16, ox0 = tx * 16;
    __syncthreads();  // previous tile's reads of the patch / pooling stage are done (weights are in place)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = tid + 256 * q;
      if (t < kTasks && !c8_ablate(a, 1)) {
        const int r = t / kGroups, g = t - r * kGroups;
        // bytes 0..11 = pixels 0..3 x (c0,c1,c2); patch column of pixel j = 4g - 1 + j (column -1 is not stored)
        const uint32_t d0 = pre[q][0], d1 = pre[q][1], d2 = pre[q][2];
        const uint32_t by[12] = {d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u, d0 >> 24, d1 & 255u, (d1 >> 8) & 255u,
                                 (d1 >> 16) & 255u, d1 >> 24, d2 & 255u, (d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24};
        uint8_t* rowp = patch + r * kC8Pitch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = 4 * g - 1 + j;
          if (col < 0) continue;
          const bool in = (pmask[q] >> j) & 1u;
          const f16x2 c01 = {(_Float16)(float)by[3 * j], (_Float16)(float)by[3 * j + 1]};
          const f16x2 c2b = {(_Float16)(float)by[3 * j + 2], (_Float16)1.0f};
          u32x2 rec = {__builtin_bit_cast(unsigned, c01), __builtin_bit_cast(unsigned, c2b)};
          if (!in) rec = (u32x2){0u, 0u};
          *reinterpret_cast<u32x2*>(rowp + col * 8) = rec;
        }
      }
    }
    __syncthreads();
    if (first_of_chunk) next_chunk = __builtin_amdgcn_readfirstlane(s_next_chunk);   // written before this tile's first barrier; scalar, so that
                                                                                      // everything derived from the tile index stays uniform
    // next tile's bytes (the first tile of the next chunk after the last one of this chunk), in flight under the MFMAs
    if (!c8_ablate(a, 8)) SERL_C8_FETCH(min(tile + 1 < t_end ? tile + 1 : next_chunk * a.chunk, a.total_tiles - 1));
    // POOL == 2: the neighbours' first column / first row (raw values written by this workgroup at earlier tiles), fetched HERE so
    // that their L2 round trip lies under the MFMAs.  Branch-free (a tile without that neighbour reads elsewhere and ignores the
    // values; `wave == 3` is a scalar branch): loads inside an exec-masked region get an `s_waitcnt vmcnt(0)` right behind them.
    float nb_col[2][4], nb_row[2][4][3];
    const bool has_right = POOL == 2 && tx + 1 < a.tiles_x, has_below = POOL == 2 && ty + 1 < a.tiles_y;
    if (POOL == 2) {
      // (a tile WITHOUT that neighbour reads the resident weights instead -- never a first_rows / first_cols slot that this
      //  workgroup is still going to write: the CU's L1 must not hold a pre-write copy of a line a later tile reads back)
      const float* dummy = reinterpret_cast<const float*>(a.whi) + li;   // >= 7168 floats; offsets below stay under 1100
      {
        const float* fcn = has_right ? a.first_cols + (((size_t)n * a.Ho + oy0 + wave * 4) * a.tiles_x + tx + 1) * 64 + li : dummy;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int i = 0; i < 4; ++i) nb_col[tn][i] = fcn[has_right ? (size_t)i * a.tiles_x * 64 + tn * 32 : (size_t)(i * 64 + tn * 32)];
      }
      if (wave == 3) {   // (uniform)
        const float* frn = has_below ? a.first_rows + (((size_t)n * a.tiles_y + ty + 1) * a.Wo + ox0) * 64 + li : dummy;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) {
            const int px = 2 * (2 * (sl >> 1) + lh) + (sl & 1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int x = min(2 * px + dx, a.Wo - 1 - ox0);   // (the clamped duplicate leaves the max unchanged)
              nb_row[tn][sl][dx] = frn[(size_t)x * 64 + tn * 32];
            }
          }
      }
    }
    int abase[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int p = wave * 64 + tm * 32 + li;
      abase[tm] = (2 * (p >> 4)) * kC8Pitch + (p & 15) * 16 + lh * 16;
    }
    const int bbase = li * (kC8WP * 2) + lh * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    if (!c8_ablate(a, 2))
#pragma unroll
    for (int ks = 0; ks < kC8K / 16; ++ks) {
      const int aoff = (ks >> 1) * kC8Pitch + (ks & 1) * 32;   // kernel row ky = ks/2, k-blocks 2(ks&1) + lh
      f16x8 apx[2], bhi[2], blo[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) apx[tm] = *reinterpret_cast<const f16x8*>(patch + abase[tm] + aoff);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int off = bbase + tn * 32 * (kC8WP * 2) + ks * 32;
        bhi[tn] = *reinterpret_cast<const f16x8*>(w_hi + off);
        blo[tn] = *reinterpret_cast<const f16x8*>(w_lo + off);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(apx[tm], blo[tn], acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(apx[tm], bhi[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    // Every load of this tile is collected HERE, before the epilogue issues its stores: gfx9 counts loads and stores in one
    // vmcnt and hipcc waits vmcnt(0) for a load whenever stores are pending too, so a load consumed after the stores (the next
    // tile's pixels at the next patch fill, the sign of gamma) exposed the stores' whole round trip once per tile.  At this point
    // the loads are one MFMA loop old; the same wait retires the PREVIOUS tile's stores (first rows / columns included) in front
    // of this tile's barriers.
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int d = 0; d < 3; ++d) asm volatile("" : : "v"(pre[q][d]), "v"(acc[1][1][15]));   // (the operand pins it behind the MFMAs)
    if (POOL == 2) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : : "v"(nb_col[tn][i]), "v"(acc[1][1][15]));
      if (wave == 3) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) asm volatile("" : : "v"(nb_row[tn][sl][dx]), "v"(acc[1][1][15]));
      }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= winv[tn];   // exact (power of two)
    if (POOL == 0) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = wave * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
          const bool ok = oy < a.Ho && ox < a.Wo;
          float* o = a.out + (((size_t)n * a.Ho + oy) * a.Wo + ox) * 64;
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const float v = ok ? acc[tm][tn][r] : 0.f;
            if (ok) o[tn * 32 + li] = v;
            s[tn] += v;
            q[tn] += v * v;
          }
        }
    } else if (c8_ablate(a, 4)) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) { const float v = acc[tm][tn][r]; s[tn] += v; q[tn] += v * v; }
    } else {  // fused 3x3/2 max-pool (every tile is full)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const float v = acc[tm][tn][r];
            s[tn] += v;
            q[tn] += v * v;
          }
      if (wave == 0) {
        float* fr = a.first_rows + (((size_t)n * a.tiles_y + (oy0 >> 4)) * a.Wo + ox0 + 4 * lh) * 64 + li;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) fr[(8 * (r >> 2) + (r & 3)) * 64 + tn * 32] = acc[0][tn][r];
      }
      if (lh == 0) {
        float* fc = a.first_cols + (((size_t)n * a.Ho + oy0 + wave * 4) * a.tiles_x + (ox0 >> 4)) * 64 + li;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              fc[(size_t)(tm * 2 + rr) * a.tiles_x * 64 + tn * 32] = acc[tm][tn][8 * rr];
      }
      // 3x3/2 max-pool of the in-tile part of every window, in registers.  A lane holds, for channel tn*32 + li, the
      // tile rows 4*wave + i (i = 0..3) and the column quads Q = 2q + lh (q = 0, 1): r = 8*(i&1) + 4q + j, tm = i>>1.
      // Horizontal: px = 2Q needs cols 4Q..4Q+2 (local), px = 2Q+1 needs cols 4Q+2, 4Q+3 and col 0 of quad Q+1, which
      // the partner lane (lane ^ 32) holds.  Vertical: py = 2*wave needs rows 0..2 (local), py = 2*wave+1 rows 2, 3 and
      // row 0 of the next wave, exchanged through LDS.  Values are sign-folded (x * sign(gamma)), so it is always a max.
      float hrow[2][4][4];   // [tn][row i][px slot = 2q + parity]
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const float sg = sgn[tn];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float part[2], v0[2];
#pragma unroll
          for (int qd = 0; qd < 2; ++qd) {
            const int rb = 8 * (i & 1) + 4 * qd;
            const float c0 = sg * acc[i >> 1][tn][rb], c1 = sg * acc[i >> 1][tn][rb + 1];
            const float c2 = sg * acc[i >> 1][tn][rb + 2], c3 = sg * acc[i >> 1][tn][rb + 3];
            hrow[tn][i][2 * qd] = fmaxf(fmaxf(c0, c1), c2);
            part[qd] = fmaxf(c2, c3);
            v0[qd] = c0;
          }
          const float r0 = __shfl_xor(v0[0], 32), r1 = __shfl_xor(v0[1], 32);
          hrow[tn][i][1] = fmaxf(part[0], lh ? r1 : r0);
          hrow[tn][i][3] = lh ? part[1] : fmaxf(part[1], r1);   // lh = 1, q = 1: column 16 belongs to the next tile
          if (POOL == 2 && has_right && lh) hrow[tn][i][3] = fmaxf(hrow[tn][i][3], sg * nb_col[tn][i]);
        }
      }
      __syncthreads();  // every wave is done reading the patch: its first 8 KB become the row-exchange buffer
      float* ex = reinterpret_cast<float*>(patch);   // [wave][tn][px slot][lane]
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) ex[((wave * 2 + tn) * 4 + sl) * 64 + lane] = hrow[tn][0][sl];
      __syncthreads();
      const int Hp = a.Ho >> 1, Wp = a.Wo >> 1;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const float sg = sgn[tn];
        float* orow = a.pooled + (((size_t)n * Hp + (oy0 >> 1) + 2 * wave) * Wp + (ox0 >> 1)) * 64 + tn * 32 + li;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          const int px = 2 * (2 * (sl >> 1) + lh) + (sl & 1);
          const float even = fmaxf(fmaxf(hrow[tn][0][sl], hrow[tn][1][sl]), hrow[tn][2][sl]);
          float odd = fmaxf(hrow[tn][2][sl], hrow[tn][3][sl]);
          if (wave < 3) odd = fmaxf(odd, ex[(((wave + 1) * 2 + tn) * 4 + sl) * 64 + lane]);
          else if (POOL == 2 && has_below)   // row 16 = the first row of the tile below
            odd = fmaxf(odd, fmaxf(fmaxf(sg * nb_row[tn][sl][0], sg * nb_row[tn][sl][1]), sg * nb_row[tn][sl][2]));
          orow[(size_t)px * 64] = sg * even;
          orow[((size_t)Wp + px) * 64] = sg * odd;
        }
      }
    }
    if (++tile == t_end) {   // last tile of the chunk (a chunk lies in one image)
      double* st = a.stats + (size_t)n * kGnGroups * 2;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) stats_flush(s[tn], q[tn], st, tn * 32 + li, 16, true);
      s[0] = s[1] = q[0] = q[1] = 0.f;
      chunk = next_chunk;
      tile = chunk * a.chunk;
      t_end = tile + a.chunk;
    }
  }
#undef SERL_C8_FETCH
#undef SERL_C8_ORDER
}

// Block-wide max of |v| (256 threads) -> power-of-two scale that puts it into [2^(top-1), 2^top).
__device__ __forceinline__ float channel_scale(float m, int top, float* red /* [4] */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  if (!(m > 0.f) || !(m < 3.0e38f)) return 1.0f;   // all-zero (or non-finite) channel
  int e;
  frexpf(m, &e);                                   // m = f * 2^e, f in [0.5, 1)
  return ldexpf(1.0f, min(max(top - e, -100), 100));
}

// conv_init weights [147][64] fp32 (k = ky*21 + kx*3 + c) -> fp16 hi / lo planes [64][224] of the folded, scaled
// weights (k' = ky*32 + kx*4 + lane; lane 3 = the bias lane, pixel slot kx = 7 is zero).  One workgroup per channel.
__global__ __launch_bounds__(256) void pack_conv_init_u8_kernel(const float* w, uint16_t* hi, uint16_t* lo, float* inv) {
  __shared__ float red[4];
  const int n = blockIdx.x, kp = threadIdx.x;
  const int ky = kp >> 5, kx = (kp >> 2) & 7, ln = kp & 3;
  const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
  double v = 0.0;
  if (kp < kC8K && kx < 7) {
    if (ln < 3) v = (double)w[(size_t)(ky * 21 + kx * 3 + ln) * 64 + n] / (255.0 * stdv[ln]);
    else
      for (int c = 0; c < 3; ++c) v -= (double)w[(size_t)(ky * 21 + kx * 3 + c) * 64 + n] * (mean[c] / stdv[c]);
  }
  const float sc = channel_scale(fabsf((float)v), 13, red);
  if (kp == 0) inv[n] = 1.0f / sc;
  if (kp >= kC8K) return;
  const float vs = (float)(v * (double)sc);
  const _Float16 h = (_Float16)clamp_h(vs);
  const _Float16 l = (_Float16)(vs - (float)h);   // unscaled residual: normal fp16 range thanks to the weight scale
  hi[(size_t)n * kC8K + kp] = __builtin_bit_cast(uint16_t, h);
  lo[(size_t)n * kC8K + kp] = __builtin_bit_cast(uint16_t, l);
}

int pack_conv_init_f16x3(const float* w, uint16_t* hi, uint16_t* lo, float* inv, hipStream_t stream) {
  static_assert(kC8K <= 256, "one thread per k'");
  hipLaunchKernelGGL(pack_conv_init_u8_kernel, dim3(64), dim3(256), 0, stream, w, hi, lo, inv);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int launch_conv_init_f16x3(const uint8_t* img, PackedConvWeights w, float* out, double* stats, int N, int H, int W,
                           int Ho, int Wo, hipStream_t stream, const float* pool_gamma, int* ticket, bool complete_pool) {
  ConvInitArgsB a{};
  a.img = img; a.whi = w.hi; a.wlo = w.lo; a.winv = w.inv; a.out = out; a.stats = stats;
  a.N = N; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.wprio = trunk_wave_prio(N);
  a.tiles_y = cdiv(Ho, 16); a.tiles_x = cdiv(Wo, 16);
  a.total_tiles = N * a.tiles_y * a.tiles_x;
  SERL_REQUIRE(ticket != nullptr, "conv_init needs a chunk ticket");
  const int tpi = a.tiles_y * a.tiles_x;
  a.chunk = (pool_gamma && complete_pool) ? tpi : (tpi % 4 == 0 ? 4 : (tpi % 2 == 0 ? 2 : 1));
  a.ticket = ticket;
#ifdef SERL_ABLATE
  { const char* e = getenv("SERL_CINIT_ABLATE"); a.ablate = e ? atoi(e) : 0; }
#endif
  // 2 persistent workgroups per CU (one per CU was measured 278 -> 377 us: issue-bound at two waves per SIMD)
  const int grid = std::min(a.total_tiles / a.chunk, 512);
  ProfScope prof("conv_init", stream);
  if (pool_gamma) {  // fused pooling: `out` (the raw_init buffer) is carved into the three compact outputs
    SERL_REQUIRE(Ho % 16 == 0 && Wo % 16 == 0, "fused conv_init pooling needs full 16x16 tiles");
    a.gamma = pool_gamma;
    a.pooled = out;
    a.first_rows = a.pooled + (size_t)N * (Ho / 2) * (Wo / 2) * 64;
    a.first_cols = a.first_rows + (size_t)N * a.tiles_y * Wo * 64;
    if (complete_pool) hipLaunchKernelGGL(conv_init_u8_kernel<2>, dim3(grid), dim3(256), kC8Lds, stream, a);
    else hipLaunchKernelGGL(conv_init_u8_kernel<1>, dim3(grid), dim3(256), kC8Lds, stream, a);
  } else {
    hipLaunchKernelGGL(conv_init_u8_kernel<0>, dim3(grid), dim3(256), kC8Lds, stream, a);
  }
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

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

// Second copy of the packed planes of a 3x3 conv in the order the row-slab kernel fetches them: per (64-channel column
// tile, group of 16 input channels, tap) one contiguous 4 KB block [plane][cout][k-half][8 halfs], which thread t of the
// 256 reads as 16 bytes at t * 16 -- fully coalesced, where the [Cout][K] planes give every pair of lanes its own row.
__global__ __launch_bounds__(256) void pack_slab_order_kernel(const uint16_t* hi, const uint16_t* lo, uint16_t* slab, int Cin, int Cout) {
  const int K = 9 * Cin, c16n = Cin >> 4;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;   // one thread per 16-byte unit
  if (e >= (long)2 * Cout * K / 8) return;
  const int t = (int)(e & 255);
  long blk = e >> 8;
  const int tap = (int)(blk % 9); blk /= 9;
  const int cg = (int)(blk % c16n);
  const int nt = (int)(blk / c16n);
  const int plane = t >> 7, cout = (t >> 1) & 63, half = t & 1;
  const uint16_t* src = (plane ? lo : hi) + (size_t)(nt * 64 + cout) * K + tap * Cin + cg * 16 + half * 8;
  reinterpret_cast<uint4*>(slab)[e] = *reinterpret_cast<const uint4*>(src);
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

int pack_slab_order_f16x3(const uint16_t* hi, const uint16_t* lo, uint16_t* slab, int Cin, int Cout, hipStream_t stream) {
  SERL_REQUIRE(Cin % 16 == 0 && Cout % 64 == 0, "slab order needs Cin %% 16 == 0 and Cout %% 64 == 0");
  const long units = (long)2 * Cout * 9 * Cin / 8;
  hipLaunchKernelGGL(pack_slab_order_kernel, dim3(cdiv(units, 256)), dim3(256), 0, stream, hi, lo, slab, Cin, Cout);
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

// Workgroups of a 2-per-CU conv kernel that can be co-resident on `stream`, counted conservatively as ONE per compute unit
// the stream may use (its CU mask if it has one; a CPX-partitioned device reports 32 CUs).  Cached per stream.
static int resident_workgroups(hipStream_t stream) {
  thread_local hipStream_t last = nullptr;
  thread_local int last_n = -1;
  if (last_n >= 0 && last == stream) return last_n;
  int dev = 0, n = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
  uint32_t mask[16] = {0};
  if (hipExtStreamGetCUMask(stream, 16, mask) == hipSuccess) {
    int m = 0;
    for (uint32_t w : mask) m += __builtin_popcount(w);
    if (m > 0 && (n == 0 || m < n)) n = m;
  } else {
    (void)hipGetLastError();
  }
  last = stream; last_n = n;
  return n;
}

struct RawInput { const float* raw; GnRef gn; };   // a conv input still in raw fp32 form + the GroupNorm to apply on load

// shapes the row-slab kernel takes: stride-1 3x3 convs with 64 or 128 output channels on 32- or 16-pixel-wide maps
static bool rowslab_shape_ok(int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksz, int stride) {
  return ksz == 3 && stride == 1 && Cout <= 128 && Cout / 64 <= kSyncPerImage && Cin % 16 == 0 && Hi == Ho && Wi == Wo &&
         (Wo == 32 || Wo == 16) && Ho % (256 / Wo) == 0 && (long)N * Hi * Wi * Cin < (1L << 31);
}
// the fused epilogue's wait needs more than 8 (G - 1) co-resident workgroups (see FuseArgs); demand twice that of the
// CUs this stream may use at ONE workgroup per CU, else run the separate elementwise pass
static bool fused_can_wait(hipStream_t stream, int G) { return resident_workgroups(stream) >= 16 * (G - 1) + 1; }

// in/out: a block's 1x1 projection offered to the launch of its conv0 (same input, stride and output shape); `done` comes back
// true when the kernel chosen for conv0 computed it as well (LDS-DMA kernel, PROJ instantiation)
struct ProjFuse { PackedConvWeights w; float* out; double* stats; bool done; };

static int launch_conv_f16x3(const char* tag, const float* in_split, PackedConvWeights w, float* out, double* stats,
                             int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int ksz, int stride,
                             hipStream_t stream, const uint8_t* zero_page = nullptr, FuseArgs* fuse = nullptr,
                             const RawInput* raw_in = nullptr, TrunkPlan::L* plan = nullptr, float* kslab = nullptr,
                             int* kctr = nullptr, ProjFuse* proj = nullptr) {
  // `fuse` (in/out): the caller's request for the fused GroupNorm epilogue (mode, gn, residual, out_split, sync, ticket);
  // on return fuse->mode is 0 when the kernel chosen for this shape cannot do it (the caller then runs the elementwise pass)
  SERL_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "conv channels unsupported (Cin %d, Cout %d)", Cin, Cout);
  ConvArgsB ab{};
  ab.wprio = trunk_wave_prio(N);
  ConvArgs& a = ab.c;
  a.in = in_split; a.w = nullptr; a.out = out; a.stats = stats;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
  a.KH = a.KW = ksz; a.stride = stride;
  a.pad = std::max((Ho - 1) * stride + ksz - Hi, 0) / 2;
  a.padw = std::max((Wo - 1) * stride + ksz - Wi, 0) / 2;
  a.M = N * Ho * Wo; a.P = Ho * Wo;
  ab.whi = w.hi; ab.wlo = w.lo; ab.winv = w.inv; ab.K = ksz * ksz * Cin;
  ab.wslab = w.slab; ab.wdma = w.dma;
  // tile configuration of the register-staged / LDS-DMA kernels: 0 = 128x128, 1 = 256x64 (Cout == 64), 4 = 128x64 with three
  // chunks in flight (fewer than 512 128x128 tiles but at least 512 128x64 ones; measured per layer at B/2, B/4, B/8),
  // 2 = 64x64 with three chunks in flight (small M: one rank's share of a data-parallel batch)
  int cfg = Cout >= 128 ? 0 : 1;
  if (cfg == 0 && (long)cdiv(a.M, 128) * (Cout / 128) < 512) cfg = 2;
  if (cfg == 2 && (long)cdiv(a.M, 128) * (Cout / 64) >= 512) cfg = 4;
  const int BM = cfg == 2 ? 64 : (cfg == 1 ? 256 : 128), BN = cfg == 0 ? 128 : 64;
  const int wrows = cfg == 2 ? 32 : 64;
  a.tiles_m = cdiv(a.M, BM); a.tiles_n = Cout / BN;
  const size_t lds = (size_t)2 * (2 * BM * 64 + 2 * BN * 64);
  // how a wave's rows relate to images (GroupNorm statistics in the epilogue): 0 = a wave lies in one image, 1 / 2 = images
  // of 32 / 16 pixels, 3 = none of these: statistics by gn_stats_kernel_b after the conv
  int pmode = (a.P % wrows == 0) ? 0 : (a.P == 32 ? 1 : (a.P == 16 ? 2 : 3));
  if (cfg == 2 && pmode == 1) pmode = 3;
  dim3 grid(a.tiles_m * a.tiles_n), block(256);
  if (plan) plan->ksplit = 1;
  {
    ProfScope prof(tag, stream);
#define SERL_LAUNCH_CONV(WM, WN, TM, TN, DEEP)                                                                                 \
  do {                                                                                                                         \
    if (pmode == 0) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<WM, WN, TM, TN, 0, DEEP>), grid, block, lds, stream, ab);        \
    else if (pmode == 1) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<WM, WN, TM, TN, 1, DEEP>), grid, block, lds, stream, ab);   \
    else if (pmode == 2) hipLaunchKernelGGL((conv_igemm_f16x3_kernel<WM, WN, TM, TN, 2, DEEP>), grid, block, lds, stream, ab);   \
    else hipLaunchKernelGGL((conv_igemm_f16x3_kernel<WM, WN, TM, TN, 3, DEEP>), grid, block, lds, stream, ab);                   \
  } while (0)
    // row-slab kernel: stride-1 3x3 convs with 64 or 128 output channels on 32- or 16-pixel-wide maps (stage 0, b1_conv1)
    const bool slab_ok = rowslab_shape_ok(N, Hi, Wi, Cin, Ho, Wo, Cout, ksz, stride) && a.pad == 1 && a.padw == 1 && w.slab != nullptr;
    SERL_REQUIRE(!raw_in || (slab_ok && Cin <= 128), "raw input is only supported by the row-slab kernel");
    // LDS-DMA kernel: everything else with at least 512 128-row tiles (32-bit byte offsets into the input)
    const bool dma_ok = (cfg == 0 || cfg == 4) && Cin % 32 == 0 && zero_page != nullptr && w.dma != nullptr &&
                        (long)N * Hi * Wi * Cin * 4 < (1L << 32);
    bool fused = false, slab_dma_used = false;
    auto can_wait = [&](int G) { return fused_can_wait(stream, G); };
    if (slab_ok) {
      a.tiles_m = a.M / 256; a.tiles_n = Cout / 64;
      if (fuse && fuse->mode && a.P % 256 == 0 && can_wait(a.P / 256 * a.tiles_n)) {
        ab.fz = *fuse; ab.fz.expected = a.P / 256; ab.fz.group = a.P / 256 * a.tiles_n; fused = true;
      }
      // 5 x s_sleep(127) ~ 20 us ~ half a tile of the stage-0 convs.  Same-call A/B (profiles/r04_ab_rs_stagger.txt): pipelined
      // step 2.5762 / 2.5747 -> 2.5523 / 2.5489 ms with 5; 3 and 8 (a quarter / three quarters of a tile) gave nothing
      static const int rs_stagger = []() { const char* e = getenv("SERL_RS_STAGGER"); return e ? atoi(e) : 5; }();
      ab.stagger = (fused && a.tiles_m * a.tiles_n >= 1024) ? rs_stagger : 0;
      // Row-major fused epilogue (rowtile_epilogue_t), default since round 5: same-call pipelined step 2.494 / 2.497 -> 2.474 / 2.471 ms,
      // serial 2.953 -> 2.936 (profiles/r05_ab_epilogue_t.txt).  SERL_EPI_T = a mask over the epilogue modes (bit mode - 1), 0 = the
      // C-layout epilogue everywhere; read per launch (the test flips it inside one process)
      { const char* e = getenv("SERL_EPI_T"); ab.epi_t = (fused && (((e ? atoi(e) : 15) >> (ab.fz.mode - 1)) & 1)) ? 1 : 0; }
      // (read per launch: the test flips it inside one process)
      const char* sd_e = getenv("SERL_SLAB_DMA");
      // LDS-DMA staging (default since round 5: pipelined step 2.432 / 2.418 -> 2.372 / 2.371 ms, serial 2.814 -> 2.770, same call):
      // conv3x3_slabdma_f16x3_kernel for split8 inputs (b0_conv1, b1_conv1), weights by DMA for the raw-input kernel (b0_conv0)
      const bool sd_on = !(sd_e && sd_e[0] == '0');
      const bool slab_dma = sd_on && !raw_in && w.dma != nullptr && zero_page != nullptr && (long)N * Hi * Wi * Cin * 4 < (1L << 32);
      if (raw_in) {
        a.in = raw_in->raw; a.in_gn = raw_in->gn;
        if (sd_on && w.dma != nullptr) {
          if (ab.epi_t) hipLaunchKernelGGL((conv3x3_rowslab_f16x3_kernel<true, true, true>), dim3(a.tiles_m * a.tiles_n), block, (size_t)kRowslabLds, stream, ab);
          else hipLaunchKernelGGL((conv3x3_rowslab_f16x3_kernel<true, true>), dim3(a.tiles_m * a.tiles_n), block, (size_t)kRowslabLds, stream, ab);
          slab_dma_used = true;
        } else
        hipLaunchKernelGGL(conv3x3_rowslab_f16x3_kernel<true>, dim3(a.tiles_m * a.tiles_n), block, (size_t)kRowslabLds, stream, ab);
      } else if (slab_dma) {
        if (ab.epi_t) hipLaunchKernelGGL(conv3x3_slabdma_f16x3_kernel<true>, dim3(a.tiles_m * a.tiles_n), block, (size_t)kSlabDmaLds, stream, ab, zero_page);
        else hipLaunchKernelGGL(conv3x3_slabdma_f16x3_kernel<false>, dim3(a.tiles_m * a.tiles_n), block, (size_t)kSlabDmaLds, stream, ab, zero_page);
        slab_dma_used = true;
      } else {
        hipLaunchKernelGGL(conv3x3_rowslab_f16x3_kernel<false>, dim3(a.tiles_m * a.tiles_n), block, (size_t)kRowslabLds, stream, ab);
      }
    } else if (dma_ok) {
      const int tn = cfg == 0 ? 2 : 1, bn = 64 * tn;
      a.tiles_m = cdiv(a.M, 128); a.tiles_n = Cout / bn;
      const dim3 g(a.tiles_m * a.tiles_n);
      const size_t l = (size_t)4 * (128 * 64 + bn * 64);   // ring of four 16-channel slots
      if (pmode == 1 && cfg == 4) pmode = 3;
      if (fuse && fuse->mode && pmode == 0 && a.P % 128 == 0 && a.tiles_n <= kSyncPerImage && can_wait(a.P / 128 * a.tiles_n)) {
        ab.fz = *fuse; ab.fz.expected = a.P / 128; ab.fz.group = a.P / 128 * a.tiles_n; fused = true;
      } else if (fuse && fuse->mode && pmode == 0 && a.P == 64 && a.M % 128 == 0 && tn == 2 && Cout / kGnGroups == 64) {
        ab.fz = *fuse; ab.fz.expected = 0; fused = true;   // LOCAL: a wave = one (image, group), no exchange
      }
#define SERL_LAUNCH_DMA(KERN, TN_, ...)                                                                              \
  do {                                                                                                              \
    if (pmode == 0) hipLaunchKernelGGL((KERN<TN_, 0 __VA_ARGS__>), g, block, l, stream, ab, zero_page, pjb);   \
    else if (pmode == 1) hipLaunchKernelGGL((KERN<TN_, 1 __VA_ARGS__>), g, block, l, stream, ab, zero_page, pjb); \
    else if (pmode == 2) hipLaunchKernelGGL((KERN<TN_, 2 __VA_ARGS__>), g, block, l, stream, ab, zero_page, pjb); \
    else hipLaunchKernelGGL((KERN<TN_, 3 __VA_ARGS__>), g, block, l, stream, ab, zero_page, pjb);              \
  } while (0)
      ConvProjB pjb{};
      const bool with_pj = proj && proj->w.dma && pmode != 3 && ksz == 3 && stride == 2 && a.pad == 0 && a.padw == 0;
      if (with_pj) {   // projection pixel == conv0's tap (0, 0): pad 0 on both axes ("SAME" padding of an even extent at stride 2)
        pjb.wdma = proj->w.dma; pjb.winv = proj->w.inv; pjb.out = proj->out; pjb.stats = proj->stats;
#define SERL_LAUNCH_DMA_PROJ(TN_)                                                                                            \
  do {                                                                                                                      \
    if (pmode == 0) hipLaunchKernelGGL((conv_dma_f16x3_kernel<TN_, 0, true>), g, block, l, stream, ab, zero_page, pjb);      \
    else if (pmode == 1) hipLaunchKernelGGL((conv_dma_f16x3_kernel<TN_, 1, true>), g, block, l, stream, ab, zero_page, pjb); \
    else hipLaunchKernelGGL((conv_dma_f16x3_kernel<TN_, 2, true>), g, block, l, stream, ab, zero_page, pjb);                 \
  } while (0)
        if (tn == 2) SERL_LAUNCH_DMA_PROJ(2);
        else SERL_LAUNCH_DMA_PROJ(1);
#undef SERL_LAUNCH_DMA_PROJ
      } else if (tn == 2) SERL_LAUNCH_DMA(conv_dma_f16x3_kernel, 2);
      else SERL_LAUNCH_DMA(conv_dma_f16x3_kernel, 1);
      if (proj) proj->done = with_pj;
#undef SERL_LAUNCH_DMA
    } else if (cfg == 0) SERL_LAUNCH_CONV(2, 2, 2, 2, 0);
    else if (cfg == 1) SERL_LAUNCH_CONV(4, 1, 2, 2, 0);
    else if (cfg == 4) SERL_LAUNCH_CONV(2, 2, 2, 1, 3);
    else {
      // small M (a rank's share of a data-parallel batch): fewer than 512 64x64 tiles leave CUs idle and every workgroup walks
      // the whole K range at global-load latency (b3_conv1 at 128 images: 256 workgroups x 144 chunks = 85-91 us).  K-SPLIT: 2 or
      // 4 workgroups per tile, partial tiles summed by the last arriver (conv_igemm_f16x3_kernel) -- no extra launch
      const int tiles = a.tiles_m * a.tiles_n, nch = ksz * ksz * (Cin >> 5);
      const char* ks_e = getenv("SERL_CONV_KSPLIT");   // (read per launch: the parity test flips it inside one process)
      const int ks_env = ks_e ? atoi(ks_e) : -1;
      int S = 1;
      // OPT-IN (SERL_CONV_KSPLIT=n >= 2: at most n workgroups per tile).  Measured at 128 / 256 images (profiles/README.md round
      // 4): two workgroups per tile speed the KERNELS up while the split launch still fits one round of the chip (b3 at 128
      // images, 256 tiles: conv0 57 -> 48 us, conv1 89 -> 61 us); with 512 tiles already (b2 at 128 images, b3 at 256) the extra
      // workgroups only queue (37 -> 51 us), four per tile never paid -- and the B/8 STEP did not move (0.765 -> 0.773 ms, two
      // same-call pairs): at that size the update chain, not the trunk stream, bounds the step.
      if (kslab && kctr && ks_env >= 2) {
        const int fit = 512;
        while (S < ks_env && tiles * S * 2 <= fit && nch % (S * 2) == 0 && nch / (S * 2) >= 8) S *= 2;
      }
      ab.ksplit = S; ab.kslab = kslab; ab.kctr = kctr;
      grid = dim3(tiles * S);
      SERL_LAUNCH_CONV(2, 2, 1, 1, 3);
      if (plan) plan->ksplit = S;
    }
    // (K-split of the small-M convs -- 2..8 workgroups per 64x64 tile, slabs reduced by the statistics kernel -- halved
    //  b3_conv1 at a per-rank batch of 32 but needed a statistics launch per conv: the step got slower; removed in round 3)
#undef SERL_LAUNCH_CONV
    if (fuse && !fused) fuse->mode = 0;
    if (plan) {
      plan->kern = slab_ok ? 'S' : (dma_ok ? 'D' : 'R');
      plan->cfg = slab_dma_used ? 9 : cfg; plan->pmode = pmode;   // (tile-config 9 = the row-slab kernel with LDS-DMA staging)
      plan->fused = fused ? (ab.fz.expected == 0 ? 2 : 1) : 0;
    }
  }
  SERL_HIP(hipGetLastError());
  if (pmode == 3) {
    hipLaunchKernelGGL(gn_stats_kernel_b, dim3(N * kGnGroups), dim3(256), 0, stream, out, stats, a.P, Cout);
    SERL_HIP(hipGetLastError());
  }
  return SERL_OK;
}

// Zeroes the statistics / arrival counters / tickets of a pass with SYSTEM-scope (write-through, sc0 sc1) 16-byte stores.
// Everything that touches these words afterwards is a memory-side atomic (stats_flush, fused_arrive_and_wait, fused_tile),
// so the zeroes must be AT the memory side too and no cache may keep a copy: a plain-store zeroing kernel (round 2, reverted
// after one unexplained parity failure) leaves the zeroed lines dirty in the L2 of whichever XCD ran the store until that
// L2 writes them back -- ordered against the next kernel only by the launch boundary's cache maintenance, i.e. outside the
// "memory-side accesses only" rule the fused epilogues rely on.  hipMemsetAsync (the blit kernel, 19 us for 1.3 MB) has the
// same property; this kernel takes ~3 us and keeps the rule by construction.
__global__ __launch_bounds__(256) void zero_sys_kernel(void* p, long n16) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256)
    __builtin_amdgcn_raw_buffer_store_b128(z, r, (int)(i * 16), 0, 17 /* sc0 | sc1 */);
}

static GnRef gn_ref_b(const double* stats, const float* gamma, const float* beta, int P, int Cc) {
  GnRef g{};
  g.stats = stats; g.gamma = gamma; g.beta = beta;
  g.inv_count = 1.0 / ((double)P * (Cc / kGnGroups));
  g.gsize = Cc / kGnGroups;
  return g;
}

// ONE fused pass at a time per process.  The fused GroupNorm epilogues WAIT for other workgroups of their launch, and their
// forward-progress argument (FuseArgs) counts on every resident workgroup that waits belonging to THIS launch.  Two agents
// whose passes run concurrently on two streams break that: the CUs can fill up with waiters of both launches while the
// workgroups they wait for cannot start -- a deadlock the spin bound turns into a trap (found by round 4's 2000-step stress
// test, which keeps a second and a third agent's passes running on other streams).  A pass therefore claims the fused path
// only if the previous fused pass was issued on the same stream (in order: no overlap) or has completed (event query);
// otherwise it runs the separate elementwise passes (same results, ~10 % slower, no waiting of any kind).  Passes of OTHER
// processes on the same GPU cannot be seen from here: run one learner process per GPU, or set SERL_GN_FUSE=0.
static std::mutex g_fused_mu;
static hipStream_t g_fused_stream = nullptr;
static hipEvent_t g_fused_done = nullptr;
static bool g_fused_any = false, g_fused_open = false;   // open: a pass issued in pieces has not issued its last piece yet
static bool claim_fused_pass(hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_fused_mu);
  if (g_fused_any && g_fused_stream != stream &&
      (g_fused_open || (g_fused_done && hipEventQuery(g_fused_done) == hipErrorNotReady))) return false;
  (void)hipGetLastError();
  g_fused_stream = stream;
  g_fused_any = g_fused_open = true;
  return true;
}
static void fused_pass_issued(hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_fused_mu);
  if (g_fused_stream != stream) return;
  g_fused_open = false;
  if (!g_fused_done && hipEventCreateWithFlags(&g_fused_done, hipEventDisableTiming) != hipSuccess) { g_fused_done = nullptr; return; }
  (void)hipEventRecord(g_fused_done, stream);
}

// Trunk forward in split-fp16 arithmetic.  Activations between kernels live in the split16 layout;
// raw conv outputs (pre-GroupNorm) and the final features stay fp32.
int trunk_forward_f16x3(const TrunkWeights& w, TrunkWorkspace& ws, TrunkPacked& pk, const uint8_t* frames, int N,
                        float* feats_out, hipStream_t stream, int stage_begin, int stage_end) {
  // [stage_begin, stage_end]: -1 = conv_init + pool, 0..3 = residual stages; a pass may be issued in consecutive pieces
  // (the intermediate activations live in the workspace), which lets the caller put an event between them
  const TrunkDims& d = ws.d;
  auto stats_of = [&](int layer) { return ws.stats + (size_t)layer * ws.max_images * kGnGroups * 2; };
  if (stage_begin < 0) {   // statistics + arrival counters + tickets
    if (ws.stats_sync_bytes % 16 == 0 && ws.stats_sync_bytes < ((size_t)1 << 31)) {
      const long n16 = (long)(ws.stats_sync_bytes / 16);
      hipLaunchKernelGGL(zero_sys_kernel, dim3((unsigned)std::min<long>(cdiv(n16, 256), 512)), dim3(256), 0, stream, (void*)ws.stats, n16);
      SERL_HIP(hipGetLastError());
    } else {
      SERL_HIP(hipMemsetAsync(ws.stats, 0, ws.stats_sync_bytes, stream));
    }
  }
  // fused epilogues for this pass?  SERL_GN_FUSE is read per pass (tests flip it inside one process); the claim is taken by the
  // piece that starts the pass and remembered in the workspace for the pieces that follow
  if (stage_begin < 0) {
    const char* e = getenv("SERL_GN_FUSE");
    ws.fuse_pass = !(e && e[0] == '0') && claim_fused_pass(stream);
  }
  const bool fuse_on = ws.fuse_pass;
  auto fuse_of = [&](int layer, int mode) {
    FuseArgs f{};
    f.mode = fuse_on ? mode : 0;
    f.sync = ws.sync + (size_t)layer * ((size_t)ws.max_images * kSyncPerImage + kSyncTickets);
    f.ticket = f.sync + (size_t)ws.max_images * kSyncPerImage;
    return f;
  };
  int rc;
  // FAST STAGE-0 INPUT: with many images conv_init completes the pooling itself (whole-image chunks) and block 0 consumes the
  // raw pooled tensor directly -- b0_conv0 applies GroupNorm + ReLU + split while staging its slabs (row-slab RAWIN), b0_conv1's
  // fused epilogue rebuilds the residual from the same raw tensor (mode 4): no elementwise pass over the pooled tensor.
  // Needs: full 16x16 conv_init tiles, at least 2 images per persistent workgroup, block 0 on the row-slab kernel with its
  // fused epilogue available.  Decided from shapes only, so that a pass issued in pieces decides the same way every time.
  const bool fuse_pool = d.h[0] % 16 == 0 && d.w[0] % 16 == 0;   // full 16 x 16 conv_init tiles: pooling fused into conv_init
  const int P0 = d.h[2] * d.w[2];
  const bool complete_pool = fuse_pool && N >= 512 && N % 512 == 0;
  const bool raw_b0 = complete_pool && fuse_on && kStageStride[0] == 1 && w.blk[0].proj == nullptr &&
                      rowslab_shape_ok(N, d.h[1], d.w[1], 64, d.h[2], d.w[2], kStageFilters[0], 3, 1) && pk.blk[0][0].slab != nullptr &&
                      pk.blk[0][1].slab != nullptr && P0 % 256 == 0 && fused_can_wait(stream, P0 / 256 * (kStageFilters[0] / 64));
  const GnRef gn_init = gn_ref_b(stats_of(0), w.gn_init_s, w.gn_init_b, d.h[0] * d.w[0], 64);
  ws.plan.images = N; ws.plan.pool = complete_pool ? 2 : (fuse_pool ? 1 : 0); ws.plan.raw_b0 = raw_b0 ? 1 : 0;
  if (stage_begin < 0) {
    if ((rc = launch_conv_init_f16x3(frames, PackedConvWeights{pk.init.hi, pk.init.lo, pk.init.inv}, ws.raw_init, stats_of(0), N, d.H,
                                     d.W, d.h[0], d.w[0], stream, fuse_pool ? w.gn_init_s : nullptr, fuse_of(0, 0).ticket, complete_pool))) return rc;
    if (raw_b0) {
      // nothing: block 0 reads ws.raw_init (the completed pooled tensor) itself
    } else if (complete_pool) {
      const long tot = (long)N * d.h[1] * d.w[1] * 16;
      ProfScope prof("gn_relu_maxpool", stream);
      hipLaunchKernelGGL(gn_relu_split_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, ws.raw_init, gn_init,
                         reinterpret_cast<uint4*>(ws.pool), N, d.h[1] * d.w[1], 64);
      SERL_HIP(hipGetLastError());
    } else if (fuse_pool) {
      const long tot = (long)N * d.h[1] * (d.w[1] / 4) * 16;   // 4 pooled pixels per thread (Wo % 16 == 0)
      const int ty = d.h[0] / 16, tx = d.w[0] / 16;
      const float* pooled = ws.raw_init;
      const float* frows = pooled + (size_t)N * d.h[1] * d.w[1] * 64;
      const float* fcols = frows + (size_t)N * ty * d.w[0] * 64;
      ProfScope prof("gn_relu_maxpool", stream);
      hipLaunchKernelGGL(pool_finish_split_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, pooled, frows, fcols, gn_init,
                         reinterpret_cast<uint4*>(ws.pool), N, d.h[0], d.w[0], ty, tx);
      SERL_HIP(hipGetLastError());
    } else {
      const long tot = (long)N * d.h[1] * d.w[1] * 16;
      ProfScope prof("gn_relu_maxpool", stream);
      hipLaunchKernelGGL(gn_relu_maxpool_split_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, ws.raw_init, gn_init,
                         reinterpret_cast<uint4*>(ws.pool), N, d.h[0], d.w[0], d.h[1], d.w[1], 64);
      SERL_HIP(hipGetLastError());
    }
  }
  static const char* kTags[kTrunkStages][3] = {{"conv_igemm/b0_conv0", "conv_igemm/b0_conv1", "conv_igemm/b0_proj"},
                                                {"conv_igemm/b1_conv0", "conv_igemm/b1_conv1", "conv_igemm/b1_proj"},
                                                {"conv_igemm/b2_conv0", "conv_igemm/b2_conv1", "conv_igemm/b2_proj"},
                                                {"conv_igemm/b3_conv0", "conv_igemm/b3_conv1", "conv_igemm/b3_proj"}};
  // One residual stage over the images [img0, img0 + nimg) of the pass.  `in_img` / `mid_img` / `out_img`: the image index at which
  // this launch sequence addresses its input tensor, its block-internal tensors (norm0, rawp, raw0, raw1) and its output tensor
  // (all 0 and nimg = N for the whole-batch pass; a sub-batch schedule can window them).  Statistics and arrival counters are
  // addressed at img0 (zeroed once per pass); a launch's tile tickets start at zero, so a sub-batch takes its own 8 ticket words (`tk`).
  auto run_stage = [&](int i, int img0, int nimg, int in_img, int mid_img, int out_img, int tk) -> int {
    const int cin = i == 0 ? 64 : kStageFilters[i - 1];
    const int f = kStageFilters[i], s = kStageStride[i];
    const int Hi = d.h[1 + i], Wi = d.w[1 + i], Ho = d.h[2 + i], Wo = d.w[2 + i], P = Ho * Wo;
    const int l0 = 1 + 3 * i, l1 = 2 + 3 * i, lp = 3 + 3 * i;
    const size_t in_px = (size_t)Hi * Wi * cin, out_px = (size_t)P * f;   // floats per image (split8 = the fp32 footprint)
    const TrunkWeights::Block& bw = w.blk[i];
    const bool has_proj = bw.proj != nullptr;
    auto st_of = [&](int layer) { return stats_of(layer) + (size_t)img0 * kGnGroups * 2; };
    auto fz_of = [&](int layer, int mode) {
      FuseArgs fz = fuse_of(layer, mode);
      fz.sync += (size_t)img0 * kSyncPerImage;
      fz.ticket += tk * 8;
      return fz;
    };
    const GnRef gn_in = gn_ref_b(stats_of(0) + (size_t)img0 * kGnGroups * 2, w.gn_init_s, w.gn_init_b, d.h[0] * d.w[0], 64);
    const float* x = (i == 0 ? ws.pool : ws.blk[i - 1].out) + (size_t)in_img * in_px;   // split16
    float* const raw0 = ws.blk[i].raw0 + (size_t)mid_img * out_px;
    float* const raw1 = ws.blk[i].raw1 + (size_t)mid_img * out_px;
    float* const rawp = ws.blk[i].rawp ? ws.blk[i].rawp + (size_t)mid_img * out_px : nullptr;
    float* const norm0 = ws.blk[i].norm0 + (size_t)mid_img * out_px;
    float* const outp = ws.blk[i].out + (size_t)out_img * out_px;
    auto pw = [&](int which) { return PackedConvWeights{pk.blk[i][which].hi, pk.blk[i][which].lo, pk.blk[i][which].inv, pk.blk[i][which].slab, pk.blk[i][which].dma}; };
    int rc;
    // GroupNorm + ReLU (+ residual) + split8 in the conv epilogue where the kernel for this shape supports it
    // (fz.mode comes back 0 otherwise and the elementwise pass below runs instead)
    FuseArgs fz0 = fz_of(l0, 1);
    fz0.gn = gn_ref_b(st_of(l0), bw.gn0_s, bw.gn0_b, P, f);
    fz0.out_split = reinterpret_cast<uint8_t*>(norm0);
    const bool raw_in = i == 0 && raw_b0;
    const RawInput rin{ws.raw_init + (size_t)in_img * in_px, gn_in};
    // FUSED PROJECTION (default; SERL_PROJ_FUSE=0 restores the separate launch): conv0's workgroups compute the block's projection
    // tile too -- no projection launch, one more pass over tap (0, 0) of an input tile that conv0 fetches anyway.  Same-call A/B
    // (profiles/r05_call1): pipelined step 2.557 -> 2.535 ms, serial 3.000 -> 2.970; tests/test_agent_gpu.py::test_fused_projection
    const char* pf_e = getenv("SERL_PROJ_FUSE");   // (read per pass: the test flips it inside one process)
    const bool proj_fuse = !(pf_e && pf_e[0] == '0');
    ProjFuse pf{pw(2), rawp, st_of(lp), false};
    if ((rc = launch_conv_f16x3(kTags[i][0], x, pw(0), raw0, st_of(l0), nimg, Hi, Wi, cin, Ho, Wo, f, 3, s, stream, pk.zero, &fz0,
                                raw_in ? &rin : nullptr, &ws.plan.conv[i][0], ws.kslab, ws.kctr, has_proj && proj_fuse ? &pf : nullptr))) return rc;
    SERL_REQUIRE(!raw_in || fz0.mode, "block 0 was planned on the fused row-slab path");
    if (has_proj && pf.done) {
      ws.plan.conv[i][2] = ws.plan.conv[i][0];
      ws.plan.conv[i][2].kern = 'F'; ws.plan.conv[i][2].fused = 0;   // 'F': rode on conv0's launch
    } else if (has_proj)
      if ((rc = launch_conv_f16x3(kTags[i][2], x, pw(2), rawp, st_of(lp), nimg, Hi, Wi, cin, Ho, Wo, f, 1, s, stream, pk.zero, nullptr,
                                  nullptr, &ws.plan.conv[i][2], ws.kslab, ws.kctr))) return rc;
    const long tot = (long)nimg * P * (f / 4);
    if (!fz0.mode) {
      ProfScope prof("gn_relu_split", stream);
      hipLaunchKernelGGL(gn_relu_split_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, raw0,
                         gn_ref_b(st_of(l0), bw.gn0_s, bw.gn0_b, P, f), reinterpret_cast<uint4*>(norm0), nimg, P, f);
      SERL_HIP(hipGetLastError());
    }
    const bool last = i == kTrunkStages - 1;
    FuseArgs fz1 = fz_of(l1, last ? 0 : (has_proj ? 3 : (raw_in ? 4 : 2)));
    fz1.gn = gn_ref_b(st_of(l1), bw.gn1_s, bw.gn1_b, P, f);
    fz1.out_split = reinterpret_cast<uint8_t*>(outp);
    if (has_proj) {
      fz1.res_raw = rawp;
      fz1.res_gn = gn_ref_b(st_of(lp), bw.gnp_s, bw.gnp_b, P, f);
    } else if (raw_in) {
      fz1.res_raw = rin.raw;
      fz1.res_gn = gn_in;
    } else {
      fz1.res_split = reinterpret_cast<const uint8_t*>(x);
    }
    if ((rc = launch_conv_f16x3(kTags[i][1], norm0, pw(1), raw1, st_of(l1), nimg, Ho, Wo, f, Ho, Wo, f, 3, 1, stream, pk.zero, &fz1,
                                nullptr, &ws.plan.conv[i][1], ws.kslab, ws.kctr))) return rc;
    SERL_REQUIRE(!raw_in || fz1.mode, "block 0 was planned on the fused row-slab path");
    if (!fz1.mode) {
      ProfScope prof("block_out", stream);
      hipLaunchKernelGGL(block_out_split_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, raw1,
                         gn_ref_b(st_of(l1), bw.gn1_s, bw.gn1_b, P, f),
                         has_proj ? nullptr : reinterpret_cast<const uint4*>(x), has_proj ? rawp : nullptr,
                         has_proj ? gn_ref_b(st_of(lp), bw.gnp_s, bw.gnp_b, P, f) : GnRef{},
                         last ? nullptr : reinterpret_cast<uint4*>(outp), last ? feats_out + (size_t)out_img * out_px : nullptr, nimg, P, f);
      SERL_HIP(hipGetLastError());
    }
    return SERL_OK;
  };
  // (A DEPTH-FIRST schedule -- stages 0 and 1 issued chunk by chunk over 128 / 256 / 512 images with every chunk-local tensor in one
  // re-used window sized for the 256 MiB Infinity Cache -- was built and measured in round 5: correct, and SLOWER in every
  // configuration (pipelined step 2.557 -> 2.638 / 2.739 / 3.030 ms at 512 / 256 / 128 images per chunk, serial 3.000 -> 3.109):
  // the kernels lose more at small M than cache-resident tensors give back.  Removed; profiles/README.md, r05_call1.)
  for (int i = 0; i < kTrunkStages; ++i) {
    if (i < stage_begin || i > stage_end) continue;
    if ((rc = run_stage(i, 0, N, 0, 0, 0, 0))) return rc;
  }
  if (fuse_on && stage_end == kTrunkStages - 1) fused_pass_issued(stream);
  return SERL_OK;
}

}  // namespace serl
