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
//
// The kernels live in headers by family (round 6): trunk_f16x3_common.h (structs, split, fused-epilogue exchange), _igemm.h
// (register-staged implicit GEMM), _dma.h (LDS-DMA ring kernel), _rowslab.h (row-slab kernels), _conv_init.h (u8 conv_init + pool),
// _elementwise.h (statistics, weight packing, split8 producers); this file keeps the kernel selection and the pass itself.
#include "trunk_f16x3_elementwise.h"

namespace serl {

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
  ab.wdma = w.dma;
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
    // row-slab kernels: stride-1 3x3 convs with 64 or 128 output channels on 32- or 16-pixel-wide maps (stage 0, b1_conv1), weights (and,
    // for a split8 input, the slab) by LDS-DMA: 32-bit byte offsets into the input
    const bool slab_ok = rowslab_shape_ok(N, Hi, Wi, Cin, Ho, Wo, Cout, ksz, stride) && a.pad == 1 && a.padw == 1 && w.dma != nullptr &&
                         (raw_in || (zero_page != nullptr && (long)N * Hi * Wi * Cin * 4 < (1L << 32)));
    SERL_REQUIRE(!raw_in || (slab_ok && Cin <= 128), "raw input is only supported by the row-slab kernel");
    // LDS-DMA kernel: everything else with at least 512 128-row tiles (32-bit byte offsets into the input)
    const bool dma_ok = (cfg == 0 || cfg == 4) && Cin % 32 == 0 && zero_page != nullptr && w.dma != nullptr &&
                        (long)N * Hi * Wi * Cin * 4 < (1L << 32);
    bool fused = false;
    auto can_wait = [&](int G) { return fused_can_wait(stream, G); };
    if (slab_ok) {
      a.tiles_m = a.M / 256; a.tiles_n = Cout / 64;
      if (fuse && fuse->mode && a.P % 256 == 0 && can_wait(a.P / 256 * a.tiles_n)) {
        ab.fz = *fuse; ab.fz.expected = a.P / 256; ab.fz.group = a.P / 256 * a.tiles_n; fused = true;
      }
      // anti-phase start: 5 x s_sleep(127) ~ 20 us ~ half a tile of the stage-0 convs.  Same-call A/B (profiles/r04_ab_rs_stagger.txt):
      // pipelined step 2.5762 / 2.5747 -> 2.5523 / 2.5489 ms with 5; 3 and 8 (a quarter / three quarters of a tile) gave nothing
      ab.stagger = (fused && a.tiles_m * a.tiles_n >= 1024) ? 5 : 0;
      // a fused launch stores ROW-major (rowtile_epilogue_t; same-call pipelined step 2.494 / 2.497 -> 2.474 / 2.471 ms against the C-layout
      // fused epilogue it replaced, profiles/r05_ab_epilogue_t.txt), an unfused one stores the raw tile
      const dim3 sg(a.tiles_m * a.tiles_n);
      if (raw_in) {   // b0_conv0 on conv_init's raw pooled output: GroupNorm + ReLU + split while the slab is staged, weights by LDS-DMA
        SERL_REQUIRE(fused, "a raw conv input is only handed over by a fused pass");
        a.in = raw_in->raw; a.in_gn = raw_in->gn;
        hipLaunchKernelGGL(conv3x3_rowslab_f16x3_kernel, sg, block, (size_t)kRowslabLds, stream, ab);
      } else {        // split8 input (b0_conv1, b1_conv1): slab and weights by LDS-DMA (same-call -2.2 % of the pipelined step against the
                      // register-staged kernel it replaced in round 5, profiles/r05_ab_slab_dma.txt)
        if (fused) hipLaunchKernelGGL(conv3x3_slabdma_f16x3_kernel<true>, sg, block, (size_t)kSlabDmaLds, stream, ab, zero_page);
        else hipLaunchKernelGGL(conv3x3_slabdma_f16x3_kernel<false>, sg, block, (size_t)kSlabDmaLds, stream, ab, zero_page);
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
      } else if (fuse && fuse->mode && pmode == 2 && a.P == 16 && a.M % 128 == 0 && tn == 2 && Cout / kGnGroups == 128) {
        ab.fz = *fuse; ab.fz.expected = 0; fused = true;   // LOCAL, stage 3: a 128 x 128 tile = eight whole images of one group (dma_tile_epilogue)
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
      plan->cfg = slab_ok ? 9 : cfg; plan->pmode = pmode;   // (tile-config 9 = the row-slab kernels, LDS-DMA staging)
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
                      rowslab_shape_ok(N, d.h[1], d.w[1], 64, d.h[2], d.w[2], kStageFilters[0], 3, 1) && pk.blk[0][0].dma != nullptr &&
                      pk.blk[0][1].dma != nullptr && P0 % 256 == 0 && fused_can_wait(stream, P0 / 256 * (kStageFilters[0] / 64));
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
    auto pw = [&](int which) { return PackedConvWeights{pk.blk[i][which].hi, pk.blk[i][which].lo, pk.blk[i][which].inv, pk.blk[i][which].dma}; };
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
    // (the last block's conv1 writes the trunk's features: plain fp32 -- out_f32 -- where its kernel has a fused epilogue for the shape: the
    //  LOCAL stage-3 form of dma_tile_epilogue since round 6; block_out below otherwise)
    FuseArgs fz1 = fz_of(l1, has_proj ? 3 : (raw_in ? 4 : 2));
    fz1.gn = gn_ref_b(st_of(l1), bw.gn1_s, bw.gn1_b, P, f);
    fz1.out_split = last ? nullptr : reinterpret_cast<uint8_t*>(outp);
    fz1.out_f32 = last ? feats_out + (size_t)out_img * out_px : nullptr;
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
