// Internal (non-ABI) interfaces between the translation units of libserl_mi355.so.
#pragma once
#include "common.h"

namespace serl {

// --------------------------------------------------------------------------------------------
// Frozen ResNet-10 trunk (reference: serl_launcher/vision/resnet_v1.py:189-286, config :383-385)
// --------------------------------------------------------------------------------------------
constexpr int kTrunkStages = 4;
constexpr int kStageFilters[kTrunkStages] = {64, 128, 256, 512};
constexpr int kStageStride[kTrunkStages] = {1, 2, 2, 2};
constexpr int kGnGroups = 4;

struct TrunkWeights {  // device pointers into the agent's parameter arena (HWIO conv kernels)
  const float* conv_init;  // [7][7][3][64]
  const float *gn_init_s, *gn_init_b;
  struct Block {
    const float *conv0, *gn0_s, *gn0_b, *conv1, *gn1_s, *gn1_b, *proj, *gnp_s, *gnp_b;
  } blk[kTrunkStages];
};

struct TrunkDims {
  int H, W;            // input image
  int h[6], w[6];      // [0]=conv_init out, [1]=pool out, [2..5]=block outputs
};
TrunkDims trunk_dims(int H, int W);

constexpr int kSyncPerImage = 8, kSyncTickets = 16;   // tickets: 8 per launch (one per XCD)
constexpr int kKsplitTiles = 1024;   // partial tiles (workgroups) a K-split conv launch may use: 16 MB of scratch
// Which kernels the last split-fp16 pass selected (introspection for the parity tests: the shapes a data-parallel rank runs
// choose other kernels than the full batch does).  kern: 'S' row-slab, 'D' LDS-DMA, 'R' register-staged implicit GEMM,
// 0 = layer absent; cfg: tile configuration of launch_conv_f16x3; fused: GroupNorm epilogue inside the conv (1 exchange, 2 local).
struct TrunkPlan {
  int images = 0;
  int pool = 0;      // 0: separate GroupNorm + max-pool pass, 1: pooled in conv_init + pool_finish, 2: completed in conv_init
  int raw_b0 = 0;    // block 0 reads conv_init's raw pooled tensor (RAWIN)
  struct L { char kern = 0; int cfg = 0, pmode = 0, fused = 0, ksplit = 1; } conv[kTrunkStages][3];
};
struct TrunkWorkspace {
  TrunkPlan plan{};
  bool fuse_pass = false;   // the pass in flight uses the fused GroupNorm epilogues (decided by its first piece)
  int max_images = 0;
  TrunkDims d{};
  float* raw_init = nullptr;  // [N][h0][w0][64]
  float* pool = nullptr;      // [N][h1][w1][64]
  struct B {
    float *raw0, *raw1, *rawp, *out, *norm0;
  } blk[kTrunkStages]{};
  // K-split scratch of the small-M conv kernel (trunk_f16x3.hip): kKsplitTiles partial 64x64 tiles + one arrival counter per
  // output tile (zero between launches: the last arriver re-zeroes its counter)
  float* kslab = nullptr;
  int* kctr = nullptr;
  double* stats = nullptr;  // 13 GN layers x [N][4][2]
  int* sync = nullptr;      // directly behind `stats` (one memset): 13 layers x (kSyncPerImage arrival counters per image + kSyncTickets ints)
  size_t stats_sync_bytes = 0;
  void* base = nullptr;     // single allocation backing everything above
  size_t bytes = 0;
};
size_t trunk_workspace_bytes(int max_images, int H, int W);
// carve `ws` out of caller-provided device memory (at least trunk_workspace_bytes big)
int trunk_workspace_bind(TrunkWorkspace& ws, void* mem, int max_images, int H, int W);
// split-fp16 copies of the 3x3 / 1x1 conv kernels (trunk_f16x3.hip): fp16 [Cout][K] hi and scaled-lo planes
struct TrunkPacked {
  struct W { uint16_t *hi, *lo; float* inv; uint16_t* dma; } blk[kTrunkStages][3]{};  // dma: copy in the piece order of the LDS-DMA / row-slab kernels, or nullptr
   // [stage][conv0, conv1, proj]; inv = per-output-channel 1/scale
  W init{};                                                   // conv_init planes
  const uint8_t* zero = nullptr;                              // 256 zero bytes (source of out-of-image taps for LDS-DMA loads)
  bool dirty = true;
};
size_t trunk_packed_bytes();
int trunk_packed_bind(TrunkPacked& p, void* mem);
// frames: u8 [n][H][W][3] (device) -> feats: f32 [n][h5][w5][512] (device).
// packed == nullptr: exact fp32 MFMA convs; otherwise the split-fp16 (f16x3) convs for the blocks.
// [stage_begin, stage_end] (split-fp16 path only): -1 = conv_init + pool, 0..3 = residual stages
int trunk_forward(const TrunkWeights& w, TrunkWorkspace& ws, const uint8_t* frames, int n,
                  float* feats_out, hipStream_t stream, TrunkPacked* packed = nullptr, int stage_begin = -1,
                  int stage_end = kTrunkStages - 1);

// --------------------------------------------------------------------------------------------
// small dense building blocks (heads.hip)
// --------------------------------------------------------------------------------------------
// rows are grouped (group g = row / rows_per_group) and every group has its own bias/gamma/beta at
// `pstride` floats apart; the pre-activation is bias + sum of S GEMM slabs laid out [g*S + s][local row][D]
struct LnFwdArgs {
  const float* slabs; int S; long slab_stride;
  const float *bias, *gamma, *beta; long pstride;
  int rows, rows_per_group;
  float* y; long ld_y; long y_goff;  // y[local_row*ld_y + group*y_goff + col] (column slices / group blocks)
  float* xhat;           // [rows][D] or nullptr
  float* rstd;           // [rows] or nullptr
  // optional fused row-dot (critic head, actor_critic_nets.py:65-73): dot_out[row] = sum_col y*dot_w + dot_b[0]
  const float* dot_w; const float* dot_b; float* dot_out;
  long dot_gstride, dot_b_gstride;  // 0: one head shared by all groups (DrQ critic); else per-group heads (ensemblized Critic)
  int relu;              // 0: tanh (the MLPs / encoder heads); 1: ReLU (BinaryClassifier, reward_classifier.py:24-26)
};
// tanh-Gaussian policy head: slabs = raw head GEMM outputs (mean, log_std); biases added here, result kept in `pre`
struct PolicyDistArgs {
  const float* slabs; int S;  // head GEMM output [mean | log_std][S K-splits][B][A]
  const float* bias_mean; const float* bias_ls; float* pre; const float* eps;
  float* act; long ld_act; float* logp; float* std_out; float* sum_logp;
  const float* lam; float* alpha_out;  // optional rider: alpha_out[0] = softplus(lam[0])
  // fused form only (GemmDesc::epi == kEpiPolicy): eps == nullptr -> N(0,1) draws hashed from (seed, global row, j), kept in eps_out
  float* eps_out; uint64_t seed; long row_offset;
  int B, A; float std_min, std_max; long slab_ld, slab_stride;
  // tf != 0 (and eps == nullptr): the draws are jax.random.normal(tf_key, (tf_rows, A))[tf_row0 + b][j] instead of the hash (jaxrng.h)
  int tf; uint32_t tf_key[2]; long tf_rows, tf_row0;
};

// Epilogues of the update chain's GEMMs ("last-arriver" fusion, heads.hip): a launch boundary between a K-split GEMM and the
// small kernel that consumed its slabs is replaced by an arrival counter per output tile -- every workgroup stores its slab
// tile write-through (sc1), drains, bumps the tile's counter, and the workgroup that draws the last ticket reads the slabs back
// (sc1 loads) IN INDEX ORDER and finishes the layer.  Same summation order as the separate kernels: bit-identical results.
enum { kEpiNone = 0, kEpiReduce = 1, kEpiPolicy = 3 };
struct GemmDesc {
  const float* A;
  const float* B;
  float* C;            // slab base; slab z at C + z*sCz
  int M, N, K;
  long sAm, sAk, sAb;  // element strides of A[m][k] and per-batch offset
  long sBk, sBn, sBb;
  long ldc, sCz;       // row stride of C and slab stride
  int nbatch, splitk;  // grid.z = nbatch*splitk, z = batch*splitk + split
  // GATHER (SmallEncoder implicit GEMM): A is the VIRTUAL im2col matrix col[row][k] of an NHWC activation tensor at A,
  //   col[row][k] = A[gtab[row] + (k / gseg) * gpitch + k % gseg]  for k < gkbias,   col[row][gkbias] = 1 (the bias column),
  // gseg = 3 * cin (one kernel row: contiguous in NHWC), gpitch = wi * cin, gtab[row] = offset of the patch's first element.
  // Forward: sAk == 1 (m = row, k = column; batch b starts at row b * M); weight gradient: sAm == 1 (m = column, k = row;
  // batch b starts at row b * K).  gseg % 16 == 0.  nullptr = ordinary strided operand.
  const int* gtab = nullptr;
  int gseg = 0, gkbias = 0;
  long gpitch = 0;
  int relu = 0;        // C = max(acc, 0) (splitk == 1 only)
  // ---- last-arriver epilogue (see above).  The slabs C are then a PADDED scratch image: ldc % 64 == 0, whole 64x64 tiles.
  int epi = kEpiNone;
  int* ctr = nullptr;        // arrival counters of this GEMM (zero before the launch; the last arriver re-zeroes its counter)
  // kEpiReduce: out[zg][m][n] = sum of the `zred` consecutive slabs z = zg*zred .. (one counter per 64x64 output tile)
  int zred = 1;
  float* out = nullptr; long ld_out = 0, out_gstride = 0;
  // kEpiPolicy: one counter per 64-row tile; the last of its nbatch * splitk workgroups samples the tanh-Gaussian
  PolicyDistArgs pd;
  GemmDesc() : A(nullptr), B(nullptr), C(nullptr), M(0), N(0), K(0), sAm(0), sAk(0), sAb(0), sBk(0), sBn(0), sBb(0), ldc(0), sCz(0),
               nbatch(0), splitk(0), pd{} {}
};
int gemm_f32(const GemmDesc& g, hipStream_t stream);

}  // namespace serl
