// Reward classifier, inference only (serl_launcher/networks/reward_classifier.py:16-113): the frozen ResNet-10 trunk
// (split-fp16 MFMA convs, trunk_f16x3.hip) -> per camera SpatialLearnedEmbeddings -> Dense -> LayerNorm -> tanh
// (vision/resnet_v1.py:81-116,324-376; common/encoding.py:26-72 with use_proprio=False) -> Dense(256) -> LayerNorm ->
// ReLU -> Dense(1).  Dropout layers are the identity at train=False, which is the only mode load_classifier_func uses.
// Same kernels as the agent's encoder heads (heads.hip); the last LayerNorm launch applies ReLU and the Dense(1) row-dot.
#include <string>
#include <vector>

#include "heads.h"
#include "internal.h"

using namespace serl;

namespace {
struct CLeaf { std::string name; long off, count; };
constexpr int kHidden = 256, kBottleneck = 256, kSleFeatures = 8;
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

struct serl_classifier {
  serl_classifier_cfg cfg{};
  int HW = 0, D = 0, E = 0;
  std::vector<CLeaf> leaves;      // trunk leaves first, then the heads; offsets into `params`
  long n_params = 0, cam_stride = 0;
  long o_sle = 0, o_dW = 0, o_db = 0, o_lng = 0, o_lnb = 0, o_w1 = 0, o_b1 = 0, o_g1 = 0, o_be1 = 0, o_w2 = 0, o_b2 = 0;
  void* arena = nullptr;
  float* params = nullptr;
  TrunkWeights tw{};
  TrunkWorkspace tws{};
  TrunkPacked tpk{};
  float *feats = nullptr, *f = nullptr, *slabs = nullptr, *enc = nullptr, *h = nullptr;
  int split0 = 1, split1 = 1;
};

namespace {

const CLeaf* find(const serl_classifier* c, const char* name) {
  for (const CLeaf& l : c->leaves)
    if (l.name == name) return &l;
  return nullptr;
}

void layout(serl_classifier* c) {
  const serl_classifier_cfg& g = c->cfg;
  const TrunkDims d = trunk_dims(g.H, g.W);
  c->HW = d.h[5] * d.w[5];
  c->D = 512 * kSleFeatures;
  c->E = kBottleneck * g.n_cam;
  long off = 0;
  auto leaf = [&](const std::string& n, long cnt) {
    c->leaves.push_back({n, off, cnt});
    const long at = off;
    off += cnt;
    return at;
  };
  leaf("trunk/conv_init", 7 * 7 * 3 * 64);
  leaf("trunk/norm_init/scale", 64);
  leaf("trunk/norm_init/bias", 64);
  int cin = 64;
  for (int i = 0; i < kTrunkStages; ++i) {
    const int f = kStageFilters[i];
    const std::string p = "trunk/block" + std::to_string(i) + "/";
    leaf(p + "conv0", 9L * cin * f); leaf(p + "gn0/scale", f); leaf(p + "gn0/bias", f);
    leaf(p + "conv1", 9L * f * f); leaf(p + "gn1/scale", f); leaf(p + "gn1/bias", f);
    if (kStageStride[i] != 1 || cin != f) { leaf(p + "proj", (long)cin * f); leaf(p + "gnp/scale", f); leaf(p + "gnp/bias", f); }
    cin = f;
  }
  long cam0 = 0;
  for (int k = 0; k < g.n_cam; ++k) {
    const std::string p = "enc/" + std::to_string(k) + "/";
    const long s = leaf(p + "sle", (long)c->HW * 512 * kSleFeatures);
    const long dW = leaf(p + "dense/kernel", (long)c->D * kBottleneck);
    const long db = leaf(p + "dense/bias", kBottleneck);
    const long lg = leaf(p + "ln/scale", kBottleneck);
    const long lb = leaf(p + "ln/bias", kBottleneck);
    if (k == 0) { cam0 = s; c->o_sle = s; c->o_dW = dW; c->o_db = db; c->o_lng = lg; c->o_lnb = lb; }
    if (k == 1) c->cam_stride = s - cam0;
  }
  if (g.n_cam == 1) c->cam_stride = off - cam0;
  c->o_w1 = leaf("head/dense0/kernel", (long)c->E * kHidden);
  c->o_b1 = leaf("head/dense0/bias", kHidden);
  c->o_g1 = leaf("head/ln/scale", kHidden);
  c->o_be1 = leaf("head/ln/bias", kHidden);
  c->o_w2 = leaf("head/dense1/kernel", kHidden);
  c->o_b2 = leaf("head/dense1/bias", 1);
  c->n_params = off;
}

int split_under(int M, int N, int groups, int smax) {   // K-split so that about 512 workgroups are in flight
  const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64) * groups;
  int s = smax;
  while (s > 1 && tiles * s > 512) s >>= 1;
  return s;
}

size_t carve(serl_classifier* c, uint8_t* base) {
  const serl_classifier_cfg& g = c->cfg;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    uint8_t* p = base ? base + off : nullptr;
    off += al256(bytes);
    return p;
  };
  const long n = g.max_batch;
  c->params = (float*)take((size_t)c->n_params * 4);
  uint8_t* pk = take(trunk_packed_bytes());
  uint8_t* ws = take(trunk_workspace_bytes(g.max_batch, g.H, g.W));
  c->feats = (float*)take((size_t)g.n_cam * n * c->HW * 512 * 4);
  c->f = (float*)take((size_t)g.n_cam * n * c->D * 4);
  c->slabs = (float*)take((size_t)32 * g.n_cam * n * kBottleneck * 4);
  c->enc = (float*)take((size_t)n * c->E * 4);
  c->h = (float*)take((size_t)n * kHidden * 4);
  if (base) {
    trunk_packed_bind(c->tpk, pk);
    trunk_workspace_bind(c->tws, ws, g.max_batch, g.H, g.W);
    auto p = [&](const std::string& nm) -> const float* {
      const CLeaf* l = find(c, nm.c_str());
      return l ? c->params + l->off : nullptr;
    };
    c->tw.conv_init = p("trunk/conv_init");
    c->tw.gn_init_s = p("trunk/norm_init/scale");
    c->tw.gn_init_b = p("trunk/norm_init/bias");
    for (int i = 0; i < kTrunkStages; ++i) {
      const std::string q = "trunk/block" + std::to_string(i) + "/";
      TrunkWeights::Block& b = c->tw.blk[i];
      b.conv0 = p(q + "conv0"); b.gn0_s = p(q + "gn0/scale"); b.gn0_b = p(q + "gn0/bias");
      b.conv1 = p(q + "conv1"); b.gn1_s = p(q + "gn1/scale"); b.gn1_b = p(q + "gn1/bias");
      b.proj = p(q + "proj"); b.gnp_s = p(q + "gnp/scale"); b.gnp_b = p(q + "gnp/bias");
    }
  }
  return off;
}

}  // namespace

extern "C" {

int serl_classifier_create(const serl_classifier_cfg* cfg, serl_classifier** out) {
  SERL_REQUIRE(cfg && out, "NULL argument");
  SERL_REQUIRE(cfg->n_cam >= 1 && cfg->n_cam <= SERL_MAX_CAMS, "n_cam %d not in [1,%d]", cfg->n_cam, SERL_MAX_CAMS);
  SERL_REQUIRE(cfg->H >= 32 && cfg->W >= 32 && cfg->max_batch >= 1, "bad classifier shape");
  SERL_HIP(hipSetDevice(cfg->device));
  serl_classifier* c = new serl_classifier();
  c->cfg = *cfg;
  layout(c);
  const size_t bytes = carve(c, nullptr);
  if (hipMalloc(&c->arena, bytes) != hipSuccess) {
    delete c;
    serl::set_error("hipMalloc of %zu bytes failed", bytes);
    return SERL_ERR_HIP;
  }
  SERL_HIP(hipMemset(c->arena, 0, bytes));
  carve(c, (uint8_t*)c->arena);
  *out = c;
  return SERL_OK;
}

int serl_classifier_destroy(serl_classifier* c) {
  if (!c) return SERL_OK;
  if (c->arena) (void)hipFree(c->arena);
  delete c;
  return SERL_OK;
}

int serl_classifier_num_leaves(serl_classifier* c) { return c ? (int)c->leaves.size() : 0; }

int serl_classifier_leaf_info(serl_classifier* c, int i, char* name_out, int name_cap, int64_t* count) {
  SERL_REQUIRE(c && i >= 0 && i < (int)c->leaves.size(), "leaf index %d out of range", i);
  const CLeaf& l = c->leaves[i];
  if (name_out && name_cap > 0) snprintf(name_out, name_cap, "%s", l.name.c_str());
  if (count) *count = l.count;
  return SERL_OK;
}

int serl_classifier_set(serl_classifier* c, const char* leaf, const float* host, int64_t count) {
  SERL_REQUIRE(c && leaf && host, "NULL argument");
  const CLeaf* l = find(c, leaf);
  SERL_REQUIRE(l, "unknown classifier leaf '%s'", leaf);
  SERL_REQUIRE(count == l->count, "leaf '%s' has %ld elements, got %ld", leaf, l->count, (long)count);
  SERL_HIP(hipSetDevice(c->cfg.device));
  SERL_HIP(hipMemcpy(c->params + l->off, host, (size_t)count * 4, hipMemcpyHostToDevice));
  if (l->name.rfind("trunk/", 0) == 0) c->tpk.dirty = true;   // fp16 planes are re-packed on the next forward
  return SERL_OK;
}

int serl_classifier_get(serl_classifier* c, const char* leaf, float* host_out, int64_t count) {
  SERL_REQUIRE(c && leaf && host_out, "NULL argument");
  const CLeaf* l = find(c, leaf);
  SERL_REQUIRE(l, "unknown classifier leaf '%s'", leaf);
  SERL_REQUIRE(count == l->count, "leaf '%s' has %ld elements, got %ld", leaf, l->count, (long)count);
  SERL_HIP(hipSetDevice(c->cfg.device));
  SERL_HIP(hipMemcpy(host_out, c->params + l->off, (size_t)count * 4, hipMemcpyDeviceToHost));
  return SERL_OK;
}

int serl_classifier_logits(serl_classifier* c, const uint8_t* dev_frames, int n, float* dev_logits, void* stream) {
  SERL_REQUIRE(c && dev_frames && dev_logits, "NULL argument");
  const serl_classifier_cfg& g = c->cfg;
  SERL_REQUIRE(n >= 1 && n <= g.max_batch, "n = %d not in [1, max_batch = %d]", n, g.max_batch);
  hipStream_t st = (hipStream_t)stream;
  SERL_HIP(hipSetDevice(g.device));
  const size_t fbytes = (size_t)g.H * g.W * 3;
  const long nmax = g.max_batch;
  for (int k = 0; k < g.n_cam; ++k) {
    int rc = trunk_forward(c->tw, c->tws, dev_frames + (size_t)k * n * fbytes, n, c->feats + (long)k * nmax * c->HW * 512, st, &c->tpk);
    if (rc) return rc;
  }
  const float* P = c->params;
  // per camera: SpatialLearnedEmbeddings -> Dense(256) (K-split GEMM) -> LayerNorm -> tanh, written side by side
  SleFwdArgs sv{c->feats, P + c->o_sle, nullptr, c->f};
  int rc = sle_fwd_multi(&sv, 1, 1.0f, n, c->HW, 512, g.n_cam, nmax * c->HW * 512, c->cam_stride, 0, nmax * c->D, st);
  if (rc) return rc;
  const int S0 = split_under(n, kBottleneck, g.n_cam, 32);
  GemmDesc g0{};
  g0.A = c->f; g0.sAm = c->D; g0.sAk = 1; g0.sAb = nmax * c->D;
  g0.B = P + c->o_dW; g0.sBk = kBottleneck; g0.sBn = 1; g0.sBb = c->cam_stride;
  g0.C = c->slabs; g0.ldc = kBottleneck; g0.sCz = (long)n * kBottleneck;
  g0.M = n; g0.N = kBottleneck; g0.K = c->D; g0.nbatch = g.n_cam; g0.splitk = S0;
  if ((rc = gemm_f32_multi(&g0, 1, st))) return rc;
  LnFwdArgs l0{};
  l0.slabs = c->slabs; l0.S = S0; l0.slab_stride = g0.sCz;
  l0.bias = P + c->o_db; l0.gamma = P + c->o_lng; l0.beta = P + c->o_lnb; l0.pstride = c->cam_stride;
  l0.rows = g.n_cam * n; l0.rows_per_group = n;
  l0.y = c->enc; l0.ld_y = c->E; l0.y_goff = kBottleneck;
  if ((rc = ln_tanh_fwd_multi(&l0, 1, kBottleneck, st))) return rc;
  // Dense(256) -> LayerNorm -> ReLU -> Dense(1)
  const int S1 = split_under(n, kHidden, 1, 8);
  GemmDesc g1{};
  g1.A = c->enc; g1.sAm = c->E; g1.sAk = 1; g1.sAb = 0;
  g1.B = P + c->o_w1; g1.sBk = kHidden; g1.sBn = 1; g1.sBb = 0;
  g1.C = c->slabs; g1.ldc = kHidden; g1.sCz = (long)n * kHidden;
  g1.M = n; g1.N = kHidden; g1.K = c->E; g1.nbatch = 1; g1.splitk = S1;
  if ((rc = gemm_f32_multi(&g1, 1, st))) return rc;
  LnFwdArgs l1{};
  l1.slabs = c->slabs; l1.S = S1; l1.slab_stride = g1.sCz;
  l1.bias = P + c->o_b1; l1.gamma = P + c->o_g1; l1.beta = P + c->o_be1; l1.pstride = 0;
  l1.rows = n; l1.rows_per_group = n;
  l1.y = c->h; l1.ld_y = kHidden; l1.y_goff = 0;
  l1.relu = 1;
  l1.dot_w = P + c->o_w2; l1.dot_b = P + c->o_b2; l1.dot_out = dev_logits;
  return ln_tanh_fwd_multi(&l1, 1, kHidden, st);
}

}  // extern "C"
