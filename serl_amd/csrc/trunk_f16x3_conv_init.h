// Part of the split-fp16 trunk (trunk_f16x3.hip includes these in order; round 6 split the 2,600-line file by kernel family):
// conv_init on raw u8 pixels with the fused 3x3/2 max-pool (conv_init_u8_kernel), its weight packing and launcher.
#pragma once
#include "trunk_f16x3_rowslab.h"

namespace serl {

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
      for (int tn = 0; tn < 2; ++tn) stats_flush(s[tn], q[tn], st, tn * 32 + li, 16, true, false);   // (read by block 0's kernels: no wait)
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

}  // namespace serl
