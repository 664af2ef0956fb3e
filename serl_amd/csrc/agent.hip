// DrQ / SAC agent: parameter arena, step orchestration and the agent half of the C ABI.
// Reference semantics (paths relative to serl_launcher/serl_launcher/):
//   agents/continuous/drq.py:255-328      update_high_utd / update_critics
//   agents/continuous/sac.py:118-299      losses, update(); :544-596 update_high_utd
//   common/common.py:124-221              3 Adam txs over the full tree, summed updates, target EMA
//   common/encoding.py:26-72              per-camera encode + proprio branch
// The frozen trunk is evaluated twice per update (augmented obs, augmented next_obs); the reference
// evaluates it a third time with target_params, whose trunk leaves equal the online ones up to
// EMA round-off (<= 1e-6 relative) -- the target copy is still maintained for export.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "heads.h"
#include "small_encoder.h"

using namespace serl;

namespace {

struct Leaf {
  std::string name;
  long off, count;
};

struct CamOff { long sle, dW, db, lng, lnb, conv; };   // sle: resnet-pretrained, conv: SmallEncoder
struct Offs {
  CamOff cam[SERL_MAX_CAMS];
  long cam_stride;
  long c_w1, c_b1, c_g1, c_be1, c_w2, c_b2, c_g2, c_be2, c_hw, c_hb;
  long p_W, p_b, p_g, p_be;
  long a_w1, a_b1, a_g1, a_be1, a_w2, a_b2, a_g2, a_be2, a_Wm, a_bm, a_Ws, a_bs;
  long lam;
  long P, Pc, Pa0, Pa1;
};

struct EncBuf {     // activations of one EncodingWrapper forward
  float* f;         // [n_cam][B][D] SLE output (after dropout)
  float* xhat;      // [n_cam*B][256]
  float* rstd;      // [n_cam*B]
  float* pxhat;     // [B][64]
  float* prstd;     // [B]
  float* enc;       // [B][ld] output slice base
  long ld;
};
struct MlpBuf {  // 2-layer LN/tanh MLP activations (ensemble: rows = E*B)
  float *h1, *xh1, *rs1, *h2, *xh2, *rs2;
};
struct CritBuf {
  float* x;  // [B][E+A]
  MlpBuf m;
  float* q;  // [ens][B]
};
struct PolBuf {
  MlpBuf m;
  float *pre, *std, *logp;  // [2][B][A], [B][A], [B]
};

constexpr int kScalars = 32;
// scalar slots (device): local sums that are all-reduced together with the gradients
enum { S_D2 = 0, S_Q = 1, S_Y = 2, S_QPI = 3, S_LOGP = 4, S_LOGP_NEXT = 5 };
// aux slots (device, rank-local, never all-reduced)
enum { X_ALPHA = 0, X_TGRAD = 1, X_NORM2 = 2 /* [2] */, X_N = 8 };
// info accumulator slots
enum { I_CL = 0, I_PQ, I_TQ, I_AL, I_TEMP, I_ENT, I_TL, I_N };
// adam_ema_kernel's info rider (heads.hip) indexes these slots by number
static_assert(S_D2 == 0 && S_Q == 1 && S_Y == 2 && S_QPI == 3 && S_LOGP == 4 && S_LOGP_NEXT == 5, "scalar slots");
static_assert(I_CL == 0 && I_PQ == 1 && I_TQ == 2 && I_AL == 3 && I_TEMP == 4 && I_ENT == 5 && I_TL == 6 && I_N <= 8, "info slots");

}  // namespace

struct serl_agent {
  serl_agent_cfg cfg{};
  int E = 0, D = 0, HW = 0, XA = 0;  // enc dim, sle dim, feature pixels, E + A
  bool state_only = false;           // n_cam == 0: SACAgent.create_states
  bool small = false;                // encoder_type == "small": trainable SmallEncoder instead of the frozen trunk
  SmallWorkspace sws{};
  Offs o{};
  std::vector<Leaf> theta_leaves, trunk_leaves;
  long trunk_count = 0;
  // device memory
  void* arena = nullptr;
  float *trunk = nullptr, *trunk_t = nullptr, *theta = nullptr, *theta_t = nullptr;
  float *m_c = nullptr, *v_c = nullptr, *m_a = nullptr, *v_a = nullptr, *m_t = nullptr, *v_t = nullptr;
  float* G = nullptr;  // [Gc (Pc) | scalars (32) | Ga (Pa1-Pa0)]
  float *Gc = nullptr, *SC = nullptr, *Ga = nullptr;
  float* info_acc = nullptr;  // [I_N]
  float* aux = nullptr;       // [X_N]
  TrunkWeights tw{};
  TrunkWorkspace tws{};
  TrunkPacked tpk{};
  int trunk_mode = 1;  // 0: exact fp32 MFMA convs, 1: split-fp16 (f16x3) convs for the blocks
  // last-arriver fusion of the update chain (heads.hip): slab reductions / the tanh-Gaussian head run inside the GEMM
  // launches that feed them, the critic loss rides on the LayerNorm backward that consumes dQ, noise is hashed where it is used.
  // SERL_CHAIN_FUSE=0 restores one launch per operation (A/B timing; the fused path is bit-identical on identical noise).
  bool fuse = true;
  // K-split budget of the chain's GEMMs (workgroups per launch): 512 by default; 256 when the caller overlaps the chain with the
  // next batch's trunk pass at a large per-rank batch (serl_agent_set_chain_budget) -- fewer workgroups queue for CU slots
  // between conv workgroups (same call: pipelined 2.605 -> 2.570 ms, conv_init next to the chain 455 -> 372 us; alone the
  // chain is 2 % slower with it, and 1024 - 2048 would be 3 - 4 % faster alone)
  long split_budget = 0;
  int* ctr = nullptr;   // arrival counters: kCtrLanes ranges of kCtrPerLane (zero between launches)
  float* feats = nullptr;  // current slot: [2][n_cam][B][HW][512]
  static constexpr int kSlots = 3;                                  // pipelined update batches in flight (see serl_mi355.h)
  float* feats_slot[kSlots + 1] = {nullptr, nullptr, nullptr, nullptr};   // 0..kSlots-1: update batches, kSlots: sample_actions
  serl_batch cur_slot[kSlots]{};
  bool slot_valid[kSlots] = {false, false, false};
  EncBuf encP{}, encT{}, encO{};
  CritBuf critT{}, crit{};
  PolBuf pol{}, polT{};
  float* slabs = nullptr; long slabs_cap = 0;  // GEMM split-K scratch (== slabs_lane[0])
  float* slabs_lane[3] = {nullptr, nullptr, nullptr};  // one per instance of a multi-instance launch
  float *dq = nullptr, *ytgt = nullptr;
  float *dh2 = nullptr, *da2 = nullptr, *dg2 = nullptr, *dh1 = nullptr, *da1 = nullptr, *dg1 = nullptr;
  float *dx = nullptr, *dz = nullptr, *dgz = nullptr, *df = nullptr, *sle_part = nullptr;
  float *dp = nullptr, *dgp = nullptr, *dpre = nullptr, *dprop_y = nullptr;
  float *eps_buf[3] = {nullptr, nullptr, nullptr};
  uint8_t* mask_buf[3] = {nullptr, nullptr, nullptr};
  float* act_tmp = nullptr;  // [B][A] sample_actions scratch
  // batch of the current update (caller-owned device memory)
  serl_batch cur{};
  bool has_batch = false;
  // The target copy of the FROZEN trunk takes an EMA step per critic update like every other leaf (common.py:124-134), but no
  // gradient or weight decay ever reaches those leaves (adamw is refused on pixel agents), so the steps are only counted here
  // and applied in one pass (frozen_ema: the same rounding sequence) when somebody reads or overwrites the trunk's target
  // leaves -- 59 MB less HBM traffic per critic step.
  int64_t trunk_ema_pending = 0;
  // optimizer bookkeeping
  int64_t step = 0;
  uint64_t noise_ctr = 0;
  serl_info last_info{};
  float lr_last[3] = {0.f, 0.f, 0.f};   // learning rates of the last apply, indexed by SERL_TX_*
  int last_global = 0;
  bool info_reset = true;
  // batch-sharded data parallelism: this rank's local batch is rows [shard_off, shard_off + local) of a global
  // batch of shard_global rows (0 = not sharded); device noise is indexed by the global row
  int64_t shard_off = 0, shard_global = 0;
  // the parameter-gradient kernels of a phase -- nothing downstream of the input-gradient chain needs them before
  // the optimizer -- are queued and issued as ONE column-sum launch and ONE grouped weight-gradient GEMM at the end
  // of the phase (flush_param_grads): 8 fewer dependent kernel boundaries per grad-step pair
  bool pg_defer = false;
  Colsum3Args pg_cs[kMaxColsum];
  int pg_ncs = 0;
  GemmDesc pg_wg[kMaxGemmGroups];
  int pg_nwg = 0;
};

namespace {

constexpr int kSleSplit = 8;
constexpr int kCtrPerLane = 4096, kCtrLanes = kMaxGemmGroups;
long pad64(long x) { return (x + 63) / 64 * 64; }
// the parameter-gradient kernels of a phase are always deferred to the end of the phase (issuing them layer by layer was
// measured 3 % slower at a per-rank batch of 32, equal at 256)

size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Bump {
  uint8_t* base;
  size_t off = 0;
  explicit Bump(void* b) : base((uint8_t*)b) {}
  template <typename T>
  T* take(size_t n) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += al(n * sizeof(T));
    return p;
  }
};

void build_layout(serl_agent* a) {
  const serl_agent_cfg& c = a->cfg;
  const TrunkDims d = c.n_cam > 0 ? trunk_dims(c.H, c.W) : TrunkDims{};
  a->state_only = c.n_cam == 0;
  a->small = !a->state_only && c.encoder_type == SERL_ENCODER_SMALL;
  a->HW = (a->state_only || a->small) ? 0 : d.h[5] * d.w[5];
  a->D = a->state_only ? 0 : (a->small ? kSmallFeat[kSmallLayers] : 512 * c.sle_features);
  a->E = a->state_only ? c.state_dim : c.bottleneck * c.n_cam + c.proprio_dim;
  a->XA = a->E + c.act_dim;
  long off = 0;
  auto leaf = [&](std::vector<Leaf>& v, const std::string& n, long cnt) {
    v.push_back({n, off, cnt});
    const long at = off;
    off += cnt;
    return at;
  };
  Offs& o = a->o;
  std::vector<Leaf>& L = a->theta_leaves;
  const long Hd = c.hidden, A = c.act_dim, N = c.ensemble;
  for (int k = 0; k < c.n_cam; ++k) {
    const std::string p = "enc/" + std::to_string(k) + "/";
    if (a->small) {   // per layer [9*cin + 1][cout]: kernel (HWIO) immediately followed by the bias (small_encoder.hip)
      o.cam[k].conv = off;
      for (int l = 0; l < kSmallLayers; ++l) {
        leaf(L, p + "conv" + std::to_string(l) + "/kernel", 9L * kSmallFeat[l] * kSmallFeat[l + 1]);
        leaf(L, p + "conv" + std::to_string(l) + "/bias", kSmallFeat[l + 1]);
      }
      o.cam[k].sle = o.cam[k].conv;
    } else {
      o.cam[k].sle = leaf(L, p + "sle", (long)a->HW * 512 * c.sle_features);
    }
    o.cam[k].dW = leaf(L, p + "dense/kernel", (long)a->D * c.bottleneck);
    o.cam[k].db = leaf(L, p + "dense/bias", c.bottleneck);
    o.cam[k].lng = leaf(L, p + "ln/scale", c.bottleneck);
    o.cam[k].lnb = leaf(L, p + "ln/bias", c.bottleneck);
  }
  o.cam_stride = c.n_cam > 1 ? o.cam[1].sle - o.cam[0].sle : off;
  o.c_w1 = leaf(L, "critic/w1", N * a->XA * Hd);
  o.c_b1 = leaf(L, "critic/b1", N * Hd);
  o.c_g1 = leaf(L, "critic/ln1/scale", N * Hd);
  o.c_be1 = leaf(L, "critic/ln1/bias", N * Hd);
  o.c_w2 = leaf(L, "critic/w2", N * Hd * Hd);
  o.c_b2 = leaf(L, "critic/b2", N * Hd);
  o.c_g2 = leaf(L, "critic/ln2/scale", N * Hd);
  o.c_be2 = leaf(L, "critic/ln2/bias", N * Hd);
  // DrQ: one Dense(1) head shared by the ensemble (drq.py:201-207); state-only SAC: ensemblize vmaps the whole
  // Critic, so every member has its own head (actor_critic_nets.py:49-73,156-164)
  o.c_hw = leaf(L, "critic/head/kernel", a->state_only ? N * Hd : Hd);
  o.c_hb = leaf(L, "critic/head/bias", a->state_only ? N : 1);
  o.Pa0 = off;
  if (!a->state_only) {
    o.p_W = leaf(L, "enc/proprio/dense/kernel", (long)c.state_dim * c.proprio_dim);
    o.p_b = leaf(L, "enc/proprio/dense/bias", c.proprio_dim);
    o.p_g = leaf(L, "enc/proprio/ln/scale", c.proprio_dim);
    o.p_be = leaf(L, "enc/proprio/ln/bias", c.proprio_dim);
  }
  o.Pc = off;
  o.a_w1 = leaf(L, "actor/w1", (long)a->E * Hd);
  o.a_b1 = leaf(L, "actor/b1", Hd);
  o.a_g1 = leaf(L, "actor/ln1/scale", Hd);
  o.a_be1 = leaf(L, "actor/ln1/bias", Hd);
  o.a_w2 = leaf(L, "actor/w2", Hd * Hd);
  o.a_b2 = leaf(L, "actor/b2", Hd);
  o.a_g2 = leaf(L, "actor/ln2/scale", Hd);
  o.a_be2 = leaf(L, "actor/ln2/bias", Hd);
  o.a_Wm = leaf(L, "actor/mean/kernel", Hd * A);
  o.a_bm = leaf(L, "actor/mean/bias", A);
  o.a_Ws = leaf(L, "actor/logstd/kernel", Hd * A);
  o.a_bs = leaf(L, "actor/logstd/bias", A);
  o.Pa1 = off;
  o.lam = leaf(L, "temp/lagrange", 1);
  o.P = off;
  // trunk
  off = 0;
  a->trunk_count = 0;
  if (a->state_only || a->small) return;   // no frozen trunk
  std::vector<Leaf>& T = a->trunk_leaves;
  leaf(T, "trunk/conv_init", 7 * 7 * 3 * 64);
  leaf(T, "trunk/norm_init/scale", 64);
  leaf(T, "trunk/norm_init/bias", 64);
  int cin = 64;
  for (int i = 0; i < kTrunkStages; ++i) {
    const int f = kStageFilters[i];
    const std::string p = "trunk/block" + std::to_string(i) + "/";
    leaf(T, p + "conv0", 9L * cin * f);
    leaf(T, p + "gn0/scale", f);
    leaf(T, p + "gn0/bias", f);
    leaf(T, p + "conv1", 9L * f * f);
    leaf(T, p + "gn1/scale", f);
    leaf(T, p + "gn1/bias", f);
    if (kStageStride[i] != 1 || cin != f) {
      leaf(T, p + "proj", (long)cin * f);
      leaf(T, p + "gnp/scale", f);
      leaf(T, p + "gnp/bias", f);
    }
    cin = f;
  }
  a->trunk_count = off;
}

const Leaf* find_leaf(const std::vector<Leaf>& v, const char* name) {
  for (const Leaf& l : v)
    if (l.name == name) return &l;
  return nullptr;
}

void bind_trunk_weights(serl_agent* a) {
  auto p = [&](const std::string& n) -> const float* {
    const Leaf* l = find_leaf(a->trunk_leaves, n.c_str());
    return l ? a->trunk + l->off : nullptr;
  };
  a->tw.conv_init = p("trunk/conv_init");
  a->tw.gn_init_s = p("trunk/norm_init/scale");
  a->tw.gn_init_b = p("trunk/norm_init/bias");
  for (int i = 0; i < kTrunkStages; ++i) {
    const std::string q = "trunk/block" + std::to_string(i) + "/";
    TrunkWeights::Block& b = a->tw.blk[i];
    b.conv0 = p(q + "conv0"); b.gn0_s = p(q + "gn0/scale"); b.gn0_b = p(q + "gn0/bias");
    b.conv1 = p(q + "conv1"); b.gn1_s = p(q + "gn1/scale"); b.gn1_b = p(q + "gn1/bias");
    b.proj = p(q + "proj"); b.gnp_s = p(q + "gnp/scale"); b.gnp_b = p(q + "gnp/bias");
  }
}

// carve everything (pass 1 with base == nullptr measures)
size_t carve(serl_agent* a, void* base) {
  const serl_agent_cfg& c = a->cfg;
  const long B = c.batch, N = c.ensemble, Hd = c.hidden, A = c.act_dim;
  const Offs& o = a->o;
  Bump b(base);
  a->trunk = b.take<float>(a->trunk_count);
  a->trunk_t = b.take<float>(a->trunk_count);
  a->theta = b.take<float>(o.P);
  a->theta_t = b.take<float>(o.P);
  a->m_c = b.take<float>(o.Pc); a->v_c = b.take<float>(o.Pc);
  a->m_a = b.take<float>(o.Pa1 - o.Pa0); a->v_a = b.take<float>(o.Pa1 - o.Pa0);
  a->m_t = b.take<float>(1); a->v_t = b.take<float>(1);
  a->G = b.take<float>(o.Pc + kScalars + (o.Pa1 - o.Pa0));
  a->Gc = a->G; a->SC = a->G ? a->G + o.Pc : nullptr; a->Ga = a->G ? a->G + o.Pc + kScalars : nullptr;
  a->info_acc = b.take<float>(8);
  a->aux = b.take<float>(X_N);
  const size_t persistent = b.off;  // zero-initialised region ends here
  for (int k = 0; k <= serl_agent::kSlots; ++k)
    a->feats_slot[k] = b.take<float>((k < serl_agent::kSlots ? 2L : 1L) * c.n_cam * B * a->HW * 512);   // (HW = 0 without a trunk)
  a->feats = a->feats_slot[0];
  auto enc = [&](EncBuf& e) {
    e.f = b.take<float>((long)c.n_cam * B * a->D);
    e.xhat = b.take<float>((long)c.n_cam * B * c.bottleneck);
    e.rstd = b.take<float>((long)c.n_cam * B);
    e.pxhat = b.take<float>(B * c.proprio_dim);
    e.prstd = b.take<float>(B);
    e.enc = nullptr; e.ld = 0;
  };
  enc(a->encP); enc(a->encT); enc(a->encO);
  float* encP_out = b.take<float>(B * a->E);
  a->encP.enc = encP_out; a->encP.ld = a->E;
  auto mlp = [&](MlpBuf& m, long rows) {
    m.h1 = b.take<float>(rows * Hd); m.xh1 = b.take<float>(rows * Hd); m.rs1 = b.take<float>(rows);
    m.h2 = b.take<float>(rows * Hd); m.xh2 = b.take<float>(rows * Hd); m.rs2 = b.take<float>(rows);
  };
  auto crit = [&](CritBuf& cb) {
    cb.x = b.take<float>(B * a->XA);
    mlp(cb.m, N * B);
    cb.q = b.take<float>(N * B);
  };
  crit(a->critT); crit(a->crit);
  a->encT.enc = a->critT.x; a->encT.ld = a->XA;
  a->encO.enc = a->crit.x; a->encO.ld = a->XA;
  for (PolBuf* pbuf : {&a->pol, &a->polT}) {
    mlp(pbuf->m, B);
    pbuf->pre = b.take<float>(2 * B * A); pbuf->std = b.take<float>(B * A); pbuf->logp = b.take<float>(B);
  }
  // (fused epilogues keep whole 64x64 slab tiles: rows and columns padded to 64)
  const long Bp = pad64(B);
  long cap = std::max<long>({(long)c.n_cam * 32 * Bp * c.bottleneck, N * Bp * pad64(a->XA), 8 * Bp * Hd, 4 * N * Bp * Hd, 16L * 64 * Hd});
  a->slabs_cap = cap;
  for (int k = 0; k < 3; ++k) a->slabs_lane[k] = b.take<float>(cap);
  a->slabs = a->slabs_lane[0];
  a->dq = b.take<float>(N * B); a->ytgt = b.take<float>(B);
  a->dh2 = b.take<float>(N * B * Hd); a->da2 = b.take<float>(N * B * Hd); a->dg2 = b.take<float>(N * B * Hd);
  a->dh1 = b.take<float>(N * B * Hd); a->da1 = b.take<float>(N * B * Hd); a->dg1 = b.take<float>(N * B * Hd);
  a->dx = b.take<float>(B * a->XA);
  a->dz = b.take<float>((long)c.n_cam * B * c.bottleneck);
  a->dgz = b.take<float>((long)c.n_cam * B * c.bottleneck);
  a->df = b.take<float>((long)c.n_cam * B * a->D);
  a->sle_part = b.take<float>((long)c.n_cam * kSleSplit * a->HW * 512 * c.sle_features);
  a->dp = b.take<float>(B * c.proprio_dim); a->dgp = b.take<float>(B * c.proprio_dim);
  a->dpre = b.take<float>(2 * B * A); a->dprop_y = b.take<float>(B * c.proprio_dim);
  for (int k = 0; k < 3; ++k) {
    a->eps_buf[k] = b.take<float>(B * A);
    a->mask_buf[k] = b.take<uint8_t>((long)c.n_cam * B * a->D);
  }
  a->act_tmp = b.take<float>(B * A);
  a->ctr = b.take<int>((long)kCtrPerLane * kCtrLanes);
  if (a->small) {
    void* smem = b.take<uint8_t>(small_workspace_bytes(c.n_cam * c.batch, c.H, c.W));
    if (base) small_workspace_bind(a->sws, smem, c.n_cam * c.batch, c.H, c.W);
  } else if (!a->state_only) {
    const int nimg = 2 * c.n_cam * c.batch;
    void* tmem = b.take<uint8_t>(trunk_workspace_bytes(nimg, c.H, c.W));
    if (base) trunk_workspace_bind(a->tws, tmem, nimg, c.H, c.W);
    void* pmem = b.take<uint8_t>(trunk_packed_bytes());
    if (base) trunk_packed_bind(a->tpk, pmem);
  }
  (void)persistent;
  return b.off;
}

#define RC(x)            \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)

// K-split of a GEMM launch: as deep as `smax` for latency when the problem is small, but never more
// workgroups than the budget -- at large per-rank batches the update chain runs beside the trunk of the
// next batch and every extra workgroup waits for a conv workgroup to retire.
// `agent_budget` is the calling agent's own serl_agent_set_chain_budget value (0 = default): no process-global state, so two
// agents (or sample_actions on another thread) never see each other's budget.
int split_for(long agent_budget, int M, int N, int groups, int smax, long budget = 0) {
  if (budget <= 0) budget = agent_budget > 0 ? agent_budget : 512L;
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64) * groups;
  int s = smax;
  while (s > 1 && tiles * s > budget) s >>= 1;
  return s;
}

// ---- EncodingWrapper forward on precomputed trunk features (encoding.py:26-72) ------------------
// Up to kMaxMulti independent instances (parameter vector x observation side x dropout mask) in four
// launches: SpatialLearnedEmbeddings, bottleneck Dense (K-split GEMM), LayerNorm+tanh, proprio branch.
// Instance i uses split-K scratch slabs_lane[i].  Samples [off, off+cnt) of the current batch.
struct EncJob {
  const float* P;        // parameter vector (online or target)
  int which;             // 0 = observations, 1 = next_observations
  const uint8_t* mask;   // dropout keep-mask, nullptr = train=False
  EncBuf* e;
  const float* act_src;  // optional rider: copy [cnt][A] actions (ld A) to act_dst (ld XA)
  float* act_dst;
  int gen_mask = 0;            // fused chain, mask == nullptr: 1 = hash the Dropout keep-mask inside the SLE kernel from mask_seed,
  uint64_t mask_seed = 0;      // 2 = jax.random.bernoulli(tf_key[camera], keep, (tf_rows, D)) rows tf_row0.. (serl_noise key_mask_*)
  const uint32_t* tf_key = nullptr; long tf_rows = 0, tf_row0 = 0;
};
int encode_multi(serl_agent* a, const EncJob* jobs, int n, int off, int cnt, hipStream_t st) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const long Bfull = a->cur.batch;
  SERL_REQUIRE(n >= 1 && n <= 3, "bad encode instance count");
  if (a->state_only) {  // encoder=None: the "encoding" is the state vector itself (+ the actions rider)
    CopyJob cj[kMaxMulti];
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const EncJob& j = jobs[i];
      cj[m++] = CopyJob{a->cur.state + ((long)j.which * Bfull + off) * c.state_dim, c.state_dim, j.e->enc, j.e->ld, c.state_dim};
      if (j.act_dst) cj[m++] = CopyJob{j.act_src, c.act_dim, j.act_dst, a->XA, c.act_dim};
    }
    return copy_cols_multi(cj, m, cnt, st);
  }
  SleFwdArgs sv[3];
  GemmDesc gd[3];
  LnFwdArgs lv[3];
  ProprioArgs pv[3];
  const int S = split_for(a->split_budget, cnt, c.bottleneck, c.n_cam * n, 32);  // K = 4096: up to 32 slices of 128
  for (int i = 0; i < n; ++i) {
    const EncJob& j = jobs[i];
    EncBuf& e = *j.e;
    sv[i] = SleFwdArgs{};
    sv[i].x = a->feats + (((long)j.which * c.n_cam) * c.batch + off) * a->HW * 512;
    sv[i].K = j.P + o.cam[0].sle;
    sv[i].mask = j.mask ? j.mask + (long)off * a->D : nullptr;
    sv[i].f = e.f;
    GemmDesc& g = gd[i];
    g = GemmDesc{};
    g.A = e.f; g.sAm = a->D; g.sAk = 1; g.sAb = (long)c.batch * a->D;
    g.B = j.P + o.cam[0].dW; g.sBk = c.bottleneck; g.sBn = 1; g.sBb = o.cam_stride;
    g.C = a->slabs_lane[i]; g.ldc = c.bottleneck; g.sCz = (long)cnt * c.bottleneck;
    g.M = cnt; g.N = c.bottleneck; g.K = a->D; g.nbatch = c.n_cam; g.splitk = S;
    LnFwdArgs& l = lv[i];
    l = LnFwdArgs{};
    l.slabs = a->slabs_lane[i]; l.S = S; l.slab_stride = g.sCz;
    l.bias = j.P + o.cam[0].db; l.gamma = j.P + o.cam[0].lng; l.beta = j.P + o.cam[0].lnb; l.pstride = o.cam_stride;
    l.rows = c.n_cam * cnt; l.rows_per_group = cnt;
    l.y = e.enc; l.ld_y = e.ld; l.y_goff = c.bottleneck;
    l.xhat = e.xhat; l.rstd = e.rstd;
    ProprioArgs& pr = pv[i];
    pr = ProprioArgs{};
    pr.state = a->cur.state + ((long)j.which * Bfull + off) * c.state_dim;
    pr.W = j.P + o.p_W; pr.b = j.P + o.p_b; pr.gamma = j.P + o.p_g; pr.beta = j.P + o.p_be;
    pr.y = e.enc + (long)c.n_cam * c.bottleneck; pr.ld_y = e.ld; pr.xhat = e.pxhat; pr.rstd = e.prstd;
    if (j.act_dst) {
      pr.copy_src = j.act_src; pr.ld_copy_src = c.act_dim; pr.copy_dst = j.act_dst; pr.ld_copy_dst = a->XA;
      pr.copy_cols = c.act_dim;
    }
  }
  if (a->fuse) {
    for (int i = 0; i < n; ++i) {
      sv[i].gen = jobs[i].gen_mask; sv[i].seed = jobs[i].mask_seed;
      sv[i].row_offset = a->shard_off + off; sv[i].rows_global = a->shard_global ? a->shard_global : Bfull;
      if (jobs[i].gen_mask == 2) {
        for (int k = 0; k < c.n_cam; ++k) { sv[i].tf_key[k][0] = jobs[i].tf_key[2 * k]; sv[i].tf_key[k][1] = jobs[i].tf_key[2 * k + 1]; }
        sv[i].tf_rows = jobs[i].tf_rows; sv[i].tf_row0 = jobs[i].tf_row0;
      }
    }
  }
  if (a->small) {   // trainable conv stack + average pool per (parameter vector, observation side); no dropout (pool "avg")
    const size_t fbytes = (size_t)c.H * c.W * 3;
    // (no Dropout on this encoder -- pool "avg" -- so two instances with the same parameters and observation side are the same
    //  pass: the actor step's critic-side and policy-side encodings of obs.  The LAST instance always runs: the backward pass
    //  reads the activations of the forward pass that ran last.)
    for (int i = 0; i < n; ++i) {
      const EncJob& j = jobs[i];
      int dup = -1;
      for (int k = i + 1; k < n; ++k)
        if (jobs[k].P == j.P && jobs[k].which == j.which) dup = k;
      if (dup >= 0) { gd[i].A = jobs[dup].e->f; continue; }
      const uint8_t* fr = a->cur.frames + (((size_t)j.which * c.n_cam) * Bfull + off) * fbytes;
      RC(small_forward(a->sws, j.P, o.cam[0].conv, o.cam_stride, fr, Bfull, c.n_cam, cnt, j.e->f, (long)c.batch * a->D, st));
    }
  } else if (a->fuse) {   // channel-blocked SLE (+ hashed Dropout mask) with the proprio branch as extra workgroups of the launch
    RC(sle_proprio_fwd_multi(sv, pv, n, 1.0f - c.dropout, cnt, a->HW, 512, c.n_cam, (long)c.batch * a->HW * 512, o.cam_stride,
                             Bfull * a->D, (long)c.batch * a->D, c.state_dim, st));
  } else {
    RC(sle_fwd_multi(sv, n, 1.0f / (1.0f - c.dropout), cnt, a->HW, 512, c.n_cam, (long)c.batch * a->HW * 512, o.cam_stride,
                     Bfull * a->D, (long)c.batch * a->D, st));
  }
  RC(gemm_f32_multi(gd, n, st));
  RC(ln_tanh_fwd_multi(lv, n, c.bottleneck, st));
  if (a->fuse && !a->small) return SERL_OK;   // (the proprio branch rode on the SLE launch)
  return proprio_fwd_multi(pv, n, c.state_dim, cnt, st);
}

// generic Dense -> LN -> tanh layer on `rows_per_group` rows for `groups` parameter groups, for n independent
// instances with identical shapes (instance i: operands of job i, split-K scratch slabs_lane[i])
struct DenseJob {
  const float* X; long ldx, x_gstride;
  const float* W; long w_gstride;
  const float *bias, *gamma, *beta; long p_gstride;
  float *y, *xhat, *rstd;
  const float *dot_w, *dot_b; float* dot_out;  // optional fused row-dot (critic head)
  long dot_gstride = 0, dot_b_gstride = 0;      // per-group heads (state-only SAC) or 0 = shared
};
int dense_ln_tanh_multi(serl_agent* a, const DenseJob* jobs, int n, int groups, int rows_per_group, int K, int splitk,
                        hipStream_t st) {
  const int Hd = a->cfg.hidden;
  SERL_REQUIRE(n >= 1 && n <= 3, "bad dense instance count");
  splitk = split_for(a->split_budget, rows_per_group, Hd, groups * n, splitk);
  GemmDesc gd[3];
  LnFwdArgs lv[3];
  for (int i = 0; i < n; ++i) {
    const DenseJob& j = jobs[i];
    GemmDesc& g = gd[i];
    g = GemmDesc{};
    g.A = j.X; g.sAm = j.ldx; g.sAk = 1; g.sAb = j.x_gstride;
    g.B = j.W; g.sBk = Hd; g.sBn = 1; g.sBb = j.w_gstride;
    g.C = a->slabs_lane[i]; g.ldc = Hd; g.sCz = (long)rows_per_group * Hd;
    g.M = rows_per_group; g.N = Hd; g.K = K; g.nbatch = groups; g.splitk = splitk;
    LnFwdArgs& l = lv[i];
    l = LnFwdArgs{};
    l.slabs = a->slabs_lane[i]; l.S = splitk; l.slab_stride = g.sCz;
    l.bias = j.bias; l.gamma = j.gamma; l.beta = j.beta; l.pstride = j.p_gstride;
    l.rows = groups * rows_per_group; l.rows_per_group = rows_per_group;
    l.y = j.y; l.ld_y = Hd; l.y_goff = (long)rows_per_group * Hd;
    l.xhat = j.xhat; l.rstd = j.rstd;
    l.dot_w = j.dot_w; l.dot_b = j.dot_b; l.dot_out = j.dot_out;
    l.dot_gstride = j.dot_gstride; l.dot_b_gstride = j.dot_b_gstride;
  }
  RC(gemm_f32_multi(gd, n, st));
  return ln_tanh_fwd_multi(lv, n, Hd, st);
}

// Policy forward + tanh-Gaussian sample (actor_critic_nets.py:179-227) for n independent inputs
struct PolJob {
  const float* P; PolBuf* pb;
  const float* enc; long ld_enc;
  const float* eps;
  float* act_out; long ld_act;
  float* sum_logp;
  float* alpha_out;  // optional rider: alpha = softplus(lagrange) of P (lagrange.py:49-50)
  // fused chain, eps == nullptr: the draws are hashed inside the head kernel from (eps_seed, global row) and kept in eps_out
  float* eps_out = nullptr; uint64_t eps_seed = 0; long eps_row0 = 0;
  const uint32_t* tf_key = nullptr; long tf_rows = 0, tf_row0 = 0;   // jax.random.normal(tf_key, (tf_rows, A)) rows tf_row0.. (serl_noise key_eps_*)
};
int policy_fwd_multi(serl_agent* a, const PolJob* jobs, int n, int cnt, hipStream_t st) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const int Hd = c.hidden, A = c.act_dim;
  SERL_REQUIRE(n >= 1 && n <= 3, "bad policy instance count");
  DenseJob d1[3], d2[3];
  GemmDesc gd[3];
  PolicyDistArgs pv[3];
  for (int i = 0; i < n; ++i) {
    const PolJob& j = jobs[i];
    const float* P = j.P;
    PolBuf& pb = *j.pb;
    d1[i] = DenseJob{j.enc, j.ld_enc, 0, P + o.a_w1, 0, P + o.a_b1, P + o.a_g1, P + o.a_be1, 0,
                     pb.m.h1, pb.m.xh1, pb.m.rs1, nullptr, nullptr, nullptr};
    d2[i] = DenseJob{pb.m.h1, Hd, 0, P + o.a_w2, 0, P + o.a_b2, P + o.a_g2, P + o.a_be2, 0,
                     pb.m.h2, pb.m.xh2, pb.m.rs2, nullptr, nullptr, nullptr};
    GemmDesc& g = gd[i];
    g = GemmDesc{};
    g.A = pb.m.h2; g.sAm = Hd; g.sAk = 1; g.sAb = 0;
    g.B = P + o.a_Wm; g.sBk = A; g.sBn = 1; g.sBb = o.a_Ws - o.a_Wm;
    g.C = a->slabs_lane[i]; g.ldc = A; g.sCz = (long)cnt * A;
    g.M = cnt; g.N = A; g.K = Hd; g.nbatch = 2; g.splitk = 4;
    pv[i] = PolicyDistArgs{};
    pv[i].slabs = a->slabs_lane[i]; pv[i].S = 4; pv[i].bias_mean = P + o.a_bm; pv[i].bias_ls = P + o.a_bs; pv[i].pre = pb.pre;
    pv[i].eps = j.eps; pv[i].act = j.act_out; pv[i].ld_act = j.ld_act; pv[i].logp = pb.logp; pv[i].std_out = pb.std;
    pv[i].sum_logp = j.sum_logp; pv[i].lam = P + o.lam; pv[i].alpha_out = j.alpha_out;
  }
  RC(dense_ln_tanh_multi(a, d1, n, 1, cnt, a->E, 8, st));
  RC(dense_ln_tanh_multi(a, d2, n, 1, cnt, Hd, 4, st));
  if (a->fuse) {   // the tanh-Gaussian head inside the head GEMM: last arriver of every 64-row tile (heads.hip, kEpiPolicy)
    SERL_REQUIRE(A <= 64, "the fused policy-head epilogue holds one 64-wide slab row per sample (act_dim %d)", A);
    SERL_REQUIRE(cdiv(cnt, 64) <= kCtrPerLane && 8 * pad64(cnt) * 64 <= a->slabs_cap, "policy head exceeds the fused epilogue's scratch");
    for (int i = 0; i < n; ++i) {
      const PolJob& j = jobs[i];
      GemmDesc& g = gd[i];
      g.ldc = 64; g.sCz = pad64(cnt) * 64;
      g.epi = kEpiPolicy; g.ctr = a->ctr + (long)i * kCtrPerLane;
      g.pd = pv[i];
      g.pd.sum_logp = nullptr;
      g.pd.eps_out = j.eps_out; g.pd.seed = j.eps_seed; g.pd.row_offset = j.eps_row0;
      g.pd.tf = j.tf_key ? 1 : 0;
      if (j.tf_key) { g.pd.tf_key[0] = j.tf_key[0]; g.pd.tf_key[1] = j.tf_key[1]; g.pd.tf_rows = j.tf_rows; g.pd.tf_row0 = j.tf_row0; }
      g.pd.B = cnt; g.pd.A = A; g.pd.std_min = c.std_min; g.pd.std_max = c.std_max;
      g.pd.slab_ld = g.ldc; g.pd.slab_stride = g.sCz;
      SERL_REQUIRE(j.eps || j.eps_out, "policy noise: neither draws nor a buffer for hashed ones");
      if (j.sum_logp) {   // sum_b log pi: a vector-sum job of the phase's deferred column-sum launch
        SERL_REQUIRE(a->pg_defer && a->pg_ncs < kMaxColsum, "no room for the log-prob sum job");
        a->pg_cs[a->pg_ncs++] = Colsum3Args{j.pb->logp, nullptr, nullptr, 1, cnt, 1, nullptr, j.sum_logp, nullptr, 0, 2};
      }
    }
    return gemm_f32_multi(gd, n, st);
  }
  RC(gemm_f32_multi(gd, n, st));
  return policy_dist_fwd_multi(pv, n, cnt, A, c.std_min, c.std_max, st);
}

// Critic ensemble forward on x = [enc | action] (actor_critic_nets.py:56-73, drq.py:201-207) for n independent
// (parameters, input) pairs; the shared Q head is fused into the LN kernel of layer 2
struct CritJob { const float* P; CritBuf* cb; };
int critic_fwd_multi(serl_agent* a, const CritJob* jobs, int n, int cnt, hipStream_t st) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const int Hd = c.hidden, N = c.ensemble;
  SERL_REQUIRE(n >= 1 && n <= 3, "bad critic instance count");
  DenseJob d1[3], d2[3];
  for (int i = 0; i < n; ++i) {
    const float* P = jobs[i].P;
    CritBuf& cb = *jobs[i].cb;
    d1[i] = DenseJob{cb.x, a->XA, 0, P + o.c_w1, (long)a->XA * Hd, P + o.c_b1, P + o.c_g1, P + o.c_be1, Hd,
                     cb.m.h1, cb.m.xh1, cb.m.rs1, nullptr, nullptr, nullptr};
    d2[i] = DenseJob{cb.m.h1, Hd, (long)cnt * Hd, P + o.c_w2, (long)Hd * Hd, P + o.c_b2, P + o.c_g2, P + o.c_be2, Hd,
                     cb.m.h2, cb.m.xh2, cb.m.rs2, P + o.c_hw, P + o.c_hb, cb.q};
    d2[i].dot_gstride = a->state_only ? Hd : 0;
    d2[i].dot_b_gstride = a->state_only ? 1 : 0;
  }
  RC(dense_ln_tanh_multi(a, d1, n, N, cnt, a->XA, 4, st));
  return dense_ln_tanh_multi(a, d2, n, N, cnt, Hd, 2, st);
}

// backward of one Dense->LN->tanh layer.  dy: [groups*rows][Hd] gradient wrt the layer output.
// Produces dpre (a->da*) and, if G != nullptr, parameter gradients into G at the given offsets
// (W grads are written directly by the GEMM: splitk == 1).
int dense_ln_tanh_bwd(serl_agent* a, const float* dy, long ld_dy, long dy_goff, const float* y, long ld_y,
                      long y_goff, const float* xhat, const float* rstd, const float* gamma, long p_gstride,
                      int groups, int rows_per_group, int D, float* dpre, float* dg, float* G, long g_off,
                      long be_off, long b_off, long pg_gstride, hipStream_t st, const float* dq = nullptr,
                      const float* dq_w = nullptr, float dq_const = 0.f, long dq_w_gstride = 0, const LossArgs* loss = nullptr) {
  LnBwdArgs l{};
  l.dy = dy; l.ld_dy = ld_dy; l.dy_goff = dy_goff;
  l.y = y; l.ld_y = ld_y; l.y_goff = y_goff;
  l.xhat = xhat; l.rstd = rstd; l.gamma = gamma; l.pstride = p_gstride;
  l.rows = groups * rows_per_group; l.rows_per_group = rows_per_group;
  l.dx = dpre; l.dg = dg;
  l.dq = dq; l.dq_w = dq_w; l.dq_const = dq_const; l.dq_w_gstride = dq_w_gstride;
  if (loss) {   // the critic loss rides on this launch and every row derives its dQ from the loss arguments (no critic_loss launch)
    l.D = D; l.dq_inline = 1;
    RC(ln_tanh_bwd_multi(&l, 1, *loss, st));
  } else {
    RC(ln_tanh_bwd(l, D, st));
  }
  if (G && a->pg_defer && a->pg_ncs < kMaxColsum) {
    a->pg_cs[a->pg_ncs++] = Colsum3Args{dg, xhat, dpre, groups, rows_per_group, D, G + g_off, G + be_off, G + b_off, pg_gstride, 0};
  } else if (G) {
    RC(colsum3(dg, xhat, dpre, groups, rows_per_group, D, G + g_off, G + be_off, G + b_off, pg_gstride, st));
  }
  return SERL_OK;
}

int flush_param_grads(serl_agent* a, hipStream_t st) {
  if (a->pg_ncs) RC(colsum3_multi(a->pg_cs, a->pg_ncs, st));
  if (a->pg_nwg) RC(gemm_f32_multi(a->pg_wg, a->pg_nwg, st));
  a->pg_ncs = a->pg_nwg = 0;
  return SERL_OK;
}

// C[g] = X[g]^T * dY[g]  (weight gradient, written directly)   X: [rows][K-dim as M], dY: [rows][N]
int wgrad(serl_agent* a, const float* X, long ldx, long x_gstride, const float* dY, long ldy, long dy_gstride, float* out,
          long ldo, long out_gstride, int groups, int Mx, int Ny, int rows, hipStream_t st) {
  GemmDesc g{};
  g.A = X; g.sAm = 1; g.sAk = ldx; g.sAb = x_gstride;
  g.B = dY; g.sBk = ldy; g.sBn = 1; g.sBb = dy_gstride;
  g.C = out; g.ldc = ldo; g.sCz = out_gstride;
  g.M = Mx; g.N = Ny; g.K = rows; g.nbatch = groups; g.splitk = 1;
  if (a->pg_defer && a->pg_nwg < kMaxGemmGroups) {
    a->pg_wg[a->pg_nwg++] = g;  // issued by flush_param_grads
    return SERL_OK;
  }
  return gemm_f32(g, st);
}

// dX[g] = dY[g] * W[g]^T   (input gradient)   W: [Kin][Nout] row-major
int igrad(const float* dY, long ldy, long dy_gstride, const float* W, long ldw, long w_gstride, float* out,
          long ldo, long out_zstride, int groups, int rows, int Kin, int Nout, hipStream_t st) {
  GemmDesc g{};
  g.A = dY; g.sAm = ldy; g.sAk = 1; g.sAb = dy_gstride;
  g.B = W; g.sBk = 1; g.sBn = ldw; g.sBb = w_gstride;
  g.C = out; g.ldc = ldo; g.sCz = out_zstride;
  g.M = rows; g.N = Kin; g.K = Nout; g.nbatch = groups; g.splitk = 1;
  return gemm_f32(g, st);
}

// dX = sum_g dY[g] * W[g]^T: the per-group products go to padded scratch slabs and the last workgroup to arrive at an output
// tile adds them in group order (heads.hip, kEpiReduce) -- igrad + reduce_slabs in one launch
int igrad_sum(serl_agent* a, const float* dY, long ldy, long dy_gstride, const float* W, long ldw, long w_gstride, float* out,
              long ldo, int groups, int rows, int Kin, int Nout, hipStream_t st) {
  GemmDesc g{};
  g.A = dY; g.sAm = ldy; g.sAk = 1; g.sAb = dy_gstride;
  g.B = W; g.sBk = 1; g.sBn = ldw; g.sBb = w_gstride;
  g.C = a->slabs; g.ldc = pad64(Kin); g.sCz = pad64(rows) * g.ldc;
  g.M = rows; g.N = Kin; g.K = Nout; g.nbatch = groups; g.splitk = 1;
  g.epi = kEpiReduce; g.zred = groups; g.ctr = a->ctr; g.out = out; g.ld_out = ldo; g.out_gstride = 0;
  SERL_REQUIRE((long)cdiv(rows, 64) * cdiv(Kin, 64) <= kCtrPerLane && groups * g.sCz <= a->slabs_cap, "input gradient exceeds the fused epilogue's scratch");
  return gemm_f32(g, st);
}

// Critic backward from dq [ens][cnt] down to dx [cnt][E+A]; parameter grads into Gc when `pg`.
// dq: [ens][cnt] gradient wrt Q, or nullptr with the constant `dq_const` for every element (actor loss)
int critic_bwd(serl_agent* a, const float* P, CritBuf& cb, int cnt, bool pg, hipStream_t st, const float* dq,
               float dq_const, const LossArgs* loss = nullptr) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const int Hd = c.hidden, N = c.ensemble;
  float* G = pg ? a->Gc : nullptr;
  if (a->fuse) {
    if (pg) {   // head kernel gradient: a deferred GEMM; a long K is split and the slabs are summed by the last arriver
      GemmDesc g{};
      const int groups = a->state_only ? N : 1, K = a->state_only ? cnt : N * cnt, split = K <= 1024 ? 1 : (a->state_only ? 4 : 8);
      g.A = a->dq; g.sAm = 1; g.sAk = 1; g.sAb = a->state_only ? cnt : 0;   // one row: dq as [1][K]
      g.B = cb.m.h2; g.sBk = Hd; g.sBn = 1; g.sBb = a->state_only ? (long)cnt * Hd : 0;
      g.M = 1; g.N = Hd; g.K = K; g.nbatch = groups; g.splitk = split;
      if (split == 1) { g.C = a->Gc + o.c_hw; g.ldc = Hd; g.sCz = Hd; }
      else {
        g.C = a->slabs; g.ldc = Hd; g.sCz = 64L * Hd;
        g.epi = kEpiReduce; g.zred = split; g.ctr = a->ctr; g.out = a->Gc + o.c_hw; g.ld_out = Hd; g.out_gstride = Hd;
        SERL_REQUIRE((long)groups * split * g.sCz <= a->slabs_cap && groups * cdiv(Hd, 64) <= kCtrPerLane, "head gradient exceeds the scratch");
      }
      SERL_REQUIRE(a->pg_defer && a->pg_nwg < kMaxGemmGroups, "no room for the deferred head-gradient GEMM");
      a->pg_wg[a->pg_nwg++] = g;
    }
    RC(dense_ln_tanh_bwd(a, nullptr, Hd, (long)cnt * Hd, cb.m.h2, Hd, (long)cnt * Hd, cb.m.xh2, cb.m.rs2, P + o.c_g2, Hd,
                         N, cnt, Hd, a->da2, a->dg2, G, o.c_g2, o.c_be2, o.c_b2, Hd, st, dq, P + o.c_hw, dq_const,
                         a->state_only ? Hd : 0, loss));
    if (pg)
      RC(wgrad(a, cb.m.h1, Hd, (long)cnt * Hd, a->da2, Hd, (long)cnt * Hd, a->Gc + o.c_w2, Hd, (long)Hd * Hd, N, Hd, Hd,
               cnt, st));
    RC(igrad(a->da2, Hd, (long)cnt * Hd, P + o.c_w2, Hd, (long)Hd * Hd, a->dh1, Hd, (long)cnt * Hd, N, cnt, Hd, Hd, st));
    RC(dense_ln_tanh_bwd(a, a->dh1, Hd, (long)cnt * Hd, cb.m.h1, Hd, (long)cnt * Hd, cb.m.xh1, cb.m.rs1, P + o.c_g1, Hd,
                         N, cnt, Hd, a->da1, a->dg1, G, o.c_g1, o.c_be1, o.c_b1, Hd, st));
    if (pg)
      RC(wgrad(a, cb.x, a->XA, 0, a->da1, Hd, (long)cnt * Hd, a->Gc + o.c_w1, Hd, (long)a->XA * Hd, N, a->XA, Hd, cnt, st));
    // dx = sum_e da1[e] * W1[e]^T, the ensemble sum inside the GEMM launch
    return igrad_sum(a, a->da1, Hd, (long)cnt * Hd, P + o.c_w1, Hd, (long)a->XA * Hd, a->dx, a->XA, N, cnt, a->XA, Hd, st);
  }
  if (pg && !a->state_only) {  // shared head kernel: dw[j] = sum_{e,b} dq*h2
    GemmDesc g{};
    g.A = a->dq; g.sAm = 0; g.sAk = 1; g.sAb = 0;
    g.B = cb.m.h2; g.sBk = Hd; g.sBn = 1; g.sBb = 0;
    const bool direct = N * cnt <= 1024;  // short K: written in place, no K-split slabs to reduce
    g.C = direct ? a->Gc + o.c_hw : a->slabs; g.ldc = Hd; g.sCz = Hd;
    g.M = 1; g.N = Hd; g.K = N * cnt; g.nbatch = 1; g.splitk = direct ? 1 : 8;
    RC(gemm_f32(g, st));
    if (!direct) RC(reduce_slabs(a->slabs, 8, Hd, 1, 1, Hd, nullptr, 0, a->Gc + o.c_hw, Hd, 0, false, st));
  } else if (pg) {  // one head per member: dw[e][j] = sum_b dq[e][b]*h2[e][b][j]
    GemmDesc g{};
    g.A = a->dq; g.sAm = 0; g.sAk = 1; g.sAb = cnt;
    g.B = cb.m.h2; g.sBk = Hd; g.sBn = 1; g.sBb = (long)cnt * Hd;
    const bool direct = cnt <= 1024;
    g.C = direct ? a->Gc + o.c_hw : a->slabs; g.ldc = Hd; g.sCz = Hd;
    g.M = 1; g.N = Hd; g.K = cnt; g.nbatch = N; g.splitk = direct ? 1 : 4;
    RC(gemm_f32(g, st));
    if (!direct) RC(reduce_slabs(a->slabs, 4, Hd, N, 1, Hd, nullptr, 0, a->Gc + o.c_hw, Hd, Hd, false, st));
  }
  // gradient through the shared head dh2 = dq (x) w is formed inside the LN backward kernel (rank-1 mode)
  RC(dense_ln_tanh_bwd(a, nullptr, Hd, (long)cnt * Hd, cb.m.h2, Hd, (long)cnt * Hd, cb.m.xh2, cb.m.rs2, P + o.c_g2, Hd,
                       N, cnt, Hd, a->da2, a->dg2, G, o.c_g2, o.c_be2, o.c_b2, Hd, st, dq, P + o.c_hw, dq_const,
                       a->state_only ? Hd : 0));
  if (pg)
    RC(wgrad(a, cb.m.h1, Hd, (long)cnt * Hd, a->da2, Hd, (long)cnt * Hd, a->Gc + o.c_w2, Hd, (long)Hd * Hd, N, Hd, Hd,
             cnt, st));
  RC(igrad(a->da2, Hd, (long)cnt * Hd, P + o.c_w2, Hd, (long)Hd * Hd, a->dh1, Hd, (long)cnt * Hd, N, cnt, Hd, Hd, st));
  RC(dense_ln_tanh_bwd(a, a->dh1, Hd, (long)cnt * Hd, cb.m.h1, Hd, (long)cnt * Hd, cb.m.xh1, cb.m.rs1, P + o.c_g1, Hd,
                       N, cnt, Hd, a->da1, a->dg1, G, o.c_g1, o.c_be1, o.c_b1, Hd, st));
  if (pg)
    RC(wgrad(a, cb.x, a->XA, 0, a->da1, Hd, (long)cnt * Hd, a->Gc + o.c_w1, Hd, (long)a->XA * Hd, N, a->XA, Hd, cnt, st));
  // dx = sum_e da1[e] * W1[e]^T
  RC(igrad(a->da1, Hd, (long)cnt * Hd, P + o.c_w1, Hd, (long)a->XA * Hd, a->slabs, a->XA, (long)cnt * a->XA, N, cnt,
           a->XA, Hd, st));
  return reduce_slabs(a->slabs, N, (long)cnt * a->XA, 1, cnt, a->XA, nullptr, 0, a->dx, a->XA, 0, false, st);
}

// Encoder-head backward for the critic path: d_enc = dx[:, :E] -> camera heads (SLE kernel, Dense,
// LayerNorm) and the proprio branch, into Gc.
int encode_bwd_critic(serl_agent* a, const float* P, EncBuf& e, int off, int cnt, hipStream_t st) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const int Bn = c.bottleneck;
  RC(dense_ln_tanh_bwd(a, a->dx, a->XA, Bn, e.enc, e.ld, Bn, e.xhat, e.rstd, P + o.cam[0].lng, o.cam_stride,
                       c.n_cam, cnt, Bn, a->dz, a->dgz, a->Gc, o.cam[0].lng, o.cam[0].lnb, o.cam[0].db,
                       o.cam_stride, st));
  RC(wgrad(a, e.f, a->D, (long)c.batch * a->D, a->dz, Bn, (long)cnt * Bn, a->Gc + o.cam[0].dW, Bn, o.cam_stride,
           c.n_cam, a->D, Bn, cnt, st));
  RC(igrad(a->dz, Bn, (long)cnt * Bn, P + o.cam[0].dW, Bn, o.cam_stride, a->df, a->D, (long)cnt * a->D, c.n_cam, cnt,
           a->D, Bn, st));
  if (a->small)   // through the average pool and the four convs of the forward pass that ran last (theta at observations)
    return small_backward(a->sws, P, o.cam[0].conv, o.cam_stride, c.n_cam, cnt, a->df, (long)cnt * a->D, a->Gc, st);
  const long sle_n = (long)a->HW * 512 * c.sle_features;
  {  // dK of every camera: one partial-sum launch (grid.z = camera x batch split) + one reduction
    const float* x = a->feats + (long)off * a->HW * 512;
    RC(sle_bwd(x, a->df, a->sle_part, cnt, a->HW, 512, kSleSplit, c.n_cam, (long)c.batch * a->HW * 512,
               (long)cnt * a->D, (long)kSleSplit * sle_n, st));
    RC(reduce_slabs(a->sle_part, kSleSplit, sle_n, c.n_cam, 1, (int)sle_n, nullptr, 0, a->Gc + o.cam[0].sle, sle_n,
                    o.cam_stride, false, st));
  }
  return SERL_OK;
}

// Fused chain: the critic path's backward below dx in one piece -- the LayerNorm backward of the camera heads (width 256) and of
// the proprio branch (width 64) share ONE launch, the SpatialLearnedEmbeddings gradient sums its batch splits itself
// (sle_bwd_fused), every parameter gradient is deferred.  `event_bucket0`: see serl_agent_critic_grads_bucketed.
int enc_proprio_bwd_critic_fused(serl_agent* a, const float* P, EncBuf& e, int off, int cnt, hipStream_t st, void* event_bucket0);

// proprio-branch backward: dy = gradient wrt the proprio code (ld/offset given), into G (Gc or Ga)
int proprio_bwd(serl_agent* a, const float* P, const float* dy, long ld_dy, const float* y, long ld_y, EncBuf& e,
                int which, int off, int cnt, float* G, long base_off, hipStream_t st) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const int Pd = c.proprio_dim;
  RC(dense_ln_tanh_bwd(a, dy, ld_dy, 0, y, ld_y, 0, e.pxhat, e.prstd, P + o.p_g, 0, 1, cnt, Pd, a->dp, a->dgp, G,
                       o.p_g - base_off, o.p_be - base_off, o.p_b - base_off, 0, st));
  const float* s = a->cur.state + ((long)which * a->cur.batch + off) * c.state_dim;
  return wgrad(a, s, c.state_dim, 0, a->dp, Pd, 0, G + (o.p_W - base_off), Pd, 0, 1, c.state_dim, Pd, cnt, st);
}

int enc_proprio_bwd_critic_fused(serl_agent* a, const float* P, EncBuf& e, int off, int cnt, hipStream_t st, void* event_bucket0) {
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  const int Bn = c.bottleneck, Pd = c.proprio_dim;
  const long pc = (long)c.n_cam * Bn;
  LnBwdArgs lb[2] = {LnBwdArgs{}, LnBwdArgs{}};
  LnBwdArgs& h = lb[0];   // camera heads
  h.dy = a->dx; h.ld_dy = a->XA; h.dy_goff = Bn;
  h.y = e.enc; h.ld_y = e.ld; h.y_goff = Bn;
  h.xhat = e.xhat; h.rstd = e.rstd; h.gamma = P + o.cam[0].lng; h.pstride = o.cam_stride;
  h.rows = c.n_cam * cnt; h.rows_per_group = cnt; h.dx = a->dz; h.dg = a->dgz; h.D = Bn;
  LnBwdArgs& q = lb[1];   // proprio branch
  q.dy = a->dx + pc; q.ld_dy = a->XA; q.dy_goff = 0;
  q.y = e.enc + pc; q.ld_y = e.ld; q.y_goff = 0;
  q.xhat = e.pxhat; q.rstd = e.prstd; q.gamma = P + o.p_g; q.pstride = 0;
  q.rows = cnt; q.rows_per_group = cnt; q.dx = a->dp; q.dg = a->dgp; q.D = Pd;
  RC(ln_tanh_bwd_multi(lb, 2, LossArgs{}, st));
  SERL_REQUIRE(a->pg_defer && a->pg_ncs + 2 <= kMaxColsum, "no room for the deferred LayerNorm gradients");
  a->pg_cs[a->pg_ncs++] = Colsum3Args{a->dgp, e.pxhat, a->dp, 1, cnt, Pd, a->Gc + o.p_g, a->Gc + o.p_be, a->Gc + o.p_b, 0, 0};
  a->pg_cs[a->pg_ncs++] = Colsum3Args{a->dgz, e.xhat, a->dz, c.n_cam, cnt, Bn, a->Gc + o.cam[0].lng, a->Gc + o.cam[0].lnb,
                                      a->Gc + o.cam[0].db, o.cam_stride, 0};
  const float* s = a->cur.state + (long)off * c.state_dim;
  RC(wgrad(a, s, c.state_dim, 0, a->dp, Pd, 0, a->Gc + o.p_W, Pd, 0, 1, c.state_dim, Pd, cnt, st));
  if (event_bucket0) {   // bucket 0 (ensemble | head | proprio | scalars) is final once these deferred jobs have run
    RC(flush_param_grads(a, st));
    SERL_HIP(hipEventRecord((hipEvent_t)event_bucket0, st));
  }
  RC(wgrad(a, e.f, a->D, (long)c.batch * a->D, a->dz, Bn, (long)cnt * Bn, a->Gc + o.cam[0].dW, Bn, o.cam_stride,
           c.n_cam, a->D, Bn, cnt, st));
  RC(igrad(a->dz, Bn, (long)cnt * Bn, P + o.cam[0].dW, Bn, o.cam_stride, a->df, a->D, (long)cnt * a->D, c.n_cam, cnt,
           a->D, Bn, st));
  if (a->small) return small_backward(a->sws, P, o.cam[0].conv, o.cam_stride, c.n_cam, cnt, a->df, (long)cnt * a->D, a->Gc, st);
  const long sle_n = (long)a->HW * 512 * c.sle_features;
  const float* x = a->feats + (long)off * a->HW * 512;
  SERL_REQUIRE((long)c.n_cam * a->HW * 2 <= kCtrPerLane, "SLE gradient exceeds the arrival counters");
  return sle_bwd_fused(x, a->df, a->sle_part, cnt, a->HW, 512, kSleSplit, c.n_cam, (long)c.batch * a->HW * 512, (long)cnt * a->D,
                       (long)kSleSplit * sle_n, a->Gc + o.cam[0].sle, o.cam_stride, a->ctr, st);
}

// Noise of one update phase: caller-provided tensors are used as they are (parity mode), missing ones are
// generated on the device -- all of them by ONE launch (NoiseBatch::flush).
struct NoiseBatch {
  NoiseJob jobs[kMaxMulti];
  int n = 0;
  int flush(hipStream_t st) { return n ? gen_noise_multi(jobs, n, st) : SERL_OK; }
};
void fetch_noise(serl_agent* a, NoiseBatch& nb, const float* given_eps, const uint8_t* given_mask, int slot,
                 int cnt_total, const float** eps, const uint8_t** mask) {
  const serl_agent_cfg& c = a->cfg;
  if (given_eps) *eps = given_eps;
  else {
    nb.jobs[nb.n++] = NoiseJob{a->eps_buf[slot], (long)cnt_total * c.act_dim, c.seed ^ (0xA5A5ull + 7919ull * (++a->noise_ctr)), 0, 0.f,
                               cnt_total, a->shard_global, a->shard_off, c.act_dim};
    *eps = a->eps_buf[slot];
  }
  if (given_mask || a->state_only || a->small) *mask = a->small ? nullptr : given_mask;  // (no dropout without the SLE branch)
  else {
    nb.jobs[nb.n++] = NoiseJob{a->mask_buf[slot], (long)c.n_cam * cnt_total * a->D,
                               c.seed ^ (0x5A5Aull + 104729ull * (++a->noise_ctr)), 1, 1.0f - c.dropout,
                               cnt_total, a->shard_global, a->shard_off, a->D};
    *mask = a->mask_buf[slot];
  }
}

// rows of the GLOBAL (mini)batch in front of this rank's `cnt` rows (serl_agent_set_shard: this rank owns rows shard_off.. of a
// global batch of shard_global; a minibatch of a high-UTD update is sharded the same way)
static long tf_row0_of(const serl_agent* a, int cnt) {
  return a->shard_global ? (a->shard_off * (long)cnt) / std::max<long>(a->cur.batch, 1) : 0;
}

// One-launch-per-operation chain (SERL_CHAIN_FUSE=0) with jax.random KEYS instead of tensors: the draws are materialised in the
// agent's noise buffers by serl_jax_fill (rows [off, off + cnt) of the buffers = rows tf_row0.. of the global arrays) and then
// taken as given tensors.  *eps / *mask are replaced only where the caller gave a key and no tensor.
int jax_noise_tensors(serl_agent* a, const uint32_t* key_eps, const uint32_t* key_mask, int slot, int off, int cnt, long global_count,
                      hipStream_t st, const float** eps, const uint8_t** mask) {
  const serl_agent_cfg& c = a->cfg;
  serl_jax_job jobs[1 + SERL_MAX_CAMS];
  int n = 0;
  const long row0 = tf_row0_of(a, cnt);
  if (key_eps && !*eps) {
    serl_jax_job& j = jobs[n++];
    j = serl_jax_job{};
    j.key[0] = key_eps[0]; j.key[1] = key_eps[1]; j.kind = SERL_JAX_NORMAL;
    j.n_total = global_count * c.act_dim; j.first = row0 * c.act_dim; j.count = (long)cnt * c.act_dim;
    j.out = a->eps_buf[slot] + (long)off * c.act_dim;
    *eps = a->eps_buf[slot];
  }
  if (key_mask && !*mask && !a->state_only && !a->small) {
    for (int k = 0; k < c.n_cam; ++k) {
      serl_jax_job& j = jobs[n++];
      j = serl_jax_job{};
      j.key[0] = key_mask[2 * k]; j.key[1] = key_mask[2 * k + 1]; j.kind = SERL_JAX_BERNOULLI_U8; j.p = 1.0f - c.dropout;
      j.n_total = global_count * a->D; j.first = row0 * a->D; j.count = (long)cnt * a->D;
      j.out = a->mask_buf[slot] + ((long)k * a->cur.batch + off) * a->D;
    }
    *mask = a->mask_buf[slot];
  }
  return n ? serl_jax_fill(c.device, jobs, n, (void*)st) : SERL_OK;
}

// Fused chain: nothing is generated ahead of time.  Missing normal draws are hashed inside the policy-head epilogue (same
// stream as gen_noise: seed and global-row indexing unchanged) and kept in eps_buf[slot]; a missing Dropout mask is hashed
// inside the SLE kernel.  The seeds advance exactly as fetch_noise advances them.
// key_eps / key_mask (serl_noise key_*, host words; nullptr = none): jax.random keys of the draws -- used where the tensor is absent.
struct FusedNoise {
  const float* eps; float* eps_out; uint64_t eps_seed; const uint8_t* mask; int gen_mask; uint64_t mask_seed;
  const uint32_t* tf_eps; const uint32_t* tf_mask;
};
FusedNoise fetch_noise_fused(serl_agent* a, const float* given_eps, const uint8_t* given_mask, int slot,
                             const uint32_t* key_eps = nullptr, const uint32_t* key_mask = nullptr) {
  const serl_agent_cfg& c = a->cfg;
  FusedNoise f{};
  f.eps = given_eps;
  f.eps_out = a->eps_buf[slot];
  if (!given_eps) {
    if (key_eps) f.tf_eps = key_eps;
    else f.eps_seed = c.seed ^ (0xA5A5ull + 7919ull * (++a->noise_ctr));
  }
  if (given_mask || a->state_only || a->small) f.mask = a->small ? nullptr : given_mask;
  else if (key_mask) { f.gen_mask = 2; f.tf_mask = key_mask; }
  else { f.gen_mask = 1; f.mask_seed = c.seed ^ (0x5A5Aull + 104729ull * (++a->noise_ctr)); }
  return f;
}


// learning rate of optimizer `tx` at `count` (optimizers.py:14-30): warm-up -> constant, or warm-up -> cosine decay
float lr_at(const serl_agent_cfg& c, int64_t count, int tx) {
  const float peak = (c.tx_lr_set[tx] || c.tx_lr[tx] > 0.f) ? c.tx_lr[tx] : c.lr;
  int warm = (tx == SERL_TX_TEMPERATURE && c.temp_warmup_steps >= 0) ? c.temp_warmup_steps : c.warmup_steps;
  if (c.tx_warmup[tx] > 0) warm = c.tx_warmup[tx] - 1;
  if (count < warm) return (float)((double)peak * (double)count / (double)warm);   // linear_schedule(0, peak, warm)
  if (c.tx_cosine_steps[tx] > 0) {   // optax.warmup_cosine_decay_schedule(0, peak, warm, decay_steps, end_value = 0)
    const double T = (double)std::max(c.tx_cosine_steps[tx] - warm, 1);
    const double k = std::min((double)(count - warm), T);
    return (float)((double)peak * 0.5 * (1.0 + std::cos(M_PI * k / T)));
  }
  return peak;
}

}  // namespace

extern "C" {

int serl_agent_create(const serl_agent_cfg* cfg, serl_agent** out) {
  SERL_REQUIRE(cfg && out, "NULL argument");
  SERL_REQUIRE(cfg->n_cam >= 0 && cfg->n_cam <= SERL_MAX_CAMS, "n_cam %d not in [0,%d]", cfg->n_cam, SERL_MAX_CAMS);
  SERL_REQUIRE(cfg->n_cam == 0 || (cfg->H >= 32 && cfg->W >= 32), "images must be at least 32x32");
  SERL_REQUIRE(cfg->encoder_type == SERL_ENCODER_RESNET_PRETRAINED || cfg->encoder_type == SERL_ENCODER_SMALL,
               "Unknown encoder type: %d", cfg->encoder_type);
  SERL_REQUIRE(cfg->n_cam > 0 || cfg->ensemble <= 16, "state-only SAC supports ensembles of at most 16");
  SERL_REQUIRE(cfg->hidden == 256 && cfg->bottleneck == 256, "hidden/bottleneck must be 256 (got %d/%d)", cfg->hidden, cfg->bottleneck);
  SERL_REQUIRE(cfg->proprio_dim == 64, "proprio_dim must be 64");
  SERL_REQUIRE(cfg->sle_features == 8, "sle_features must be 8");
  SERL_REQUIRE(cfg->batch >= 1 && cfg->ensemble >= 2 && cfg->state_dim >= 1 && cfg->act_dim >= 1 && cfg->act_dim <= 64, "bad dims");
  for (int t = 0; t < 3; ++t) {
    // adamw decays EVERY leaf of the tree the optimizer is given (common.py:142-147 passes the full params), i.e. it
    // would un-freeze the pretrained trunk and make the online and target trunks differ, which the two-pass trunk
    // evaluation relies on being equal: supported for the state-only agent only
    if (cfg->tx_weight_decay_on[t] && cfg->n_cam > 0) {
      set_error("weight_decay on a pixel agent would decay the frozen pretrained trunk (optimizers.py:39-42 over the full tree): unsupported");
      return SERL_ERR_UNSUPPORTED;
    }
    SERL_REQUIRE(cfg->tx_lr[t] >= 0.f && cfg->tx_warmup[t] >= 0 && cfg->tx_cosine_steps[t] >= 0 && cfg->tx_clip_norm[t] >= 0.f,
                 "bad optimizer options for optimizer %d", t);
  }
  SERL_HIP(hipSetDevice(cfg->device));
  serl_agent* a = new serl_agent();
  a->cfg = *cfg;
  { const char* e = getenv("SERL_CHAIN_FUSE"); a->fuse = !(e && e[0] == '0'); }
  build_layout(a);
  const size_t bytes = carve(a, nullptr);
  hipError_t e = hipMalloc(&a->arena, bytes);
  if (e != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the agent arena failed: %s", bytes, hipGetErrorString(e));
    delete a;
    return SERL_ERR_HIP;
  }
  carve(a, a->arena);
  SERL_HIP(hipMemset(a->arena, 0, bytes));
  if (!a->state_only && !a->small) bind_trunk_weights(a);
  *out = a;
  return SERL_OK;
}

int serl_agent_destroy(serl_agent* a) {
  if (!a) return SERL_OK;
  (void)hipSetDevice(a->cfg.device);
  (void)hipDeviceSynchronize();
  if (a->arena) (void)hipFree(a->arena);
  delete a;
  return SERL_OK;
}

int serl_agent_num_leaves(serl_agent* a) {
  return a ? (int)(a->theta_leaves.size() + a->trunk_leaves.size()) : -1;
}

int serl_agent_leaf_info(serl_agent* a, int i, char* name_out, int name_cap, int64_t* count) {
  SERL_REQUIRE(a && name_out && count, "NULL argument");
  const int nt = (int)a->theta_leaves.size();
  SERL_REQUIRE(i >= 0 && i < serl_agent_num_leaves(a), "leaf index %d out of range", i);
  const Leaf& l = i < nt ? a->theta_leaves[i] : a->trunk_leaves[i - nt];
  snprintf(name_out, name_cap, "%s", l.name.c_str());
  *count = l.count;
  return SERL_OK;
}

static int flush_trunk_ema(serl_agent* a) {
  if (a->trunk_ema_pending > 0 && a->trunk_count > 0) {
    RC(frozen_ema(a->trunk, a->trunk_t, a->trunk_count, a->cfg.tau, a->trunk_ema_pending, nullptr));
    SERL_HIP(hipDeviceSynchronize());
  }
  a->trunk_ema_pending = 0;
  return SERL_OK;
}

// resolves (section, leaf) -> device pointer (nullptr = outside that optimizer's support)
static int resolve(serl_agent* a, const char* section, const char* leaf, float** ptr, long* count, bool* zero) {
  *zero = false;
  const Leaf* t = find_leaf(a->trunk_leaves, leaf);
  const Leaf* l = t ? t : find_leaf(a->theta_leaves, leaf);
  SERL_REQUIRE(l != nullptr, "unknown leaf '%s'", leaf);
  *count = l->count;
  const std::string s(section);
  const Offs& o = a->o;
  if (s == "params") { *ptr = (t ? a->trunk : a->theta) + l->off; return SERL_OK; }
  if (s == "target_params") { *ptr = (t ? a->trunk_t : a->theta_t) + l->off; return SERL_OK; }
  const bool mu = s.size() > 3 && s.substr(s.size() - 3) == "/mu";
  const bool nu = s.size() > 3 && s.substr(s.size() - 3) == "/nu";
  SERL_REQUIRE(s.rfind("opt/", 0) == 0 && (mu || nu), "unknown section '%s'", section);
  const std::string tx = s.substr(4, s.size() - 7);
  *ptr = nullptr;
  *zero = true;  // exact zeros outside the support (the gradient there is always exactly zero)
  if (t) return SERL_OK;
  if (tx == "critic") {
    if (l->off < o.Pc) { *ptr = (mu ? a->m_c : a->v_c) + l->off; *zero = false; }
  } else if (tx == "actor") {
    if (l->off >= o.Pa0 && l->off < o.Pa1) { *ptr = (mu ? a->m_a : a->v_a) + (l->off - o.Pa0); *zero = false; }
  } else if (tx == "temperature") {
    if (l->off == o.lam) { *ptr = mu ? a->m_t : a->v_t; *zero = false; }
  } else {
    set_error("unknown optimizer '%s'", tx.c_str());
    return SERL_ERR_INVALID;
  }
  return SERL_OK;
}

int serl_agent_set(serl_agent* a, const char* section, const char* leaf, const float* host, int64_t count) {
  SERL_REQUIRE(a && section && leaf && host, "NULL argument");
  SERL_HIP(hipSetDevice(a->cfg.device));
  float* p; long n; bool zero;
  int rc = resolve(a, section, leaf, &p, &n, &zero);
  if (rc) return rc;
  SERL_REQUIRE(n == count, "leaf '%s' has %ld elements, got %lld", leaf, n, (long long)count);
  if (std::strncmp(leaf, "trunk/", 6) == 0) { SERL_HIP(hipDeviceSynchronize()); RC(flush_trunk_ema(a)); }   // pending EMA steps belong to the old values
  if (zero) {
    for (long i = 0; i < n; ++i)
      SERL_REQUIRE(host[i] == 0.0f, "'%s' of '%s' lies outside the optimizer's support and must be zero", section, leaf);
    return SERL_OK;
  }
  SERL_HIP(hipMemcpy(p, host, sizeof(float) * n, hipMemcpyHostToDevice));
  if (std::strncmp(leaf, "trunk/", 6) == 0) a->tpk.dirty = true;
  return SERL_OK;
}

int serl_agent_get(serl_agent* a, const char* section, const char* leaf, float* host_out, int64_t count) {
  SERL_REQUIRE(a && section && leaf && host_out, "NULL argument");
  SERL_HIP(hipSetDevice(a->cfg.device));
  float* p; long n; bool zero;
  int rc = resolve(a, section, leaf, &p, &n, &zero);
  if (rc) return rc;
  SERL_REQUIRE(n == count, "leaf '%s' has %ld elements, got %lld", leaf, n, (long long)count);
  if (zero) { std::memset(host_out, 0, sizeof(float) * n); return SERL_OK; }
  SERL_HIP(hipDeviceSynchronize());
  if (std::strncmp(leaf, "trunk/", 6) == 0 && std::strcmp(section, "target_params") == 0) RC(flush_trunk_ema(a));
  SERL_HIP(hipMemcpy(host_out, p, sizeof(float) * n, hipMemcpyDeviceToHost));
  return SERL_OK;
}

int serl_agent_set_trunk_mode(serl_agent* a, int mode) {
  SERL_REQUIRE(a && (mode == 0 || mode == 1), "trunk mode must be 0 (fp32) or 1 (f16x3)");
  a->trunk_mode = mode;
  return SERL_OK;
}

int serl_agent_set_chain_budget(serl_agent* a, int workgroups) {
  SERL_REQUIRE(a && workgroups >= 0 && workgroups <= 8192, "bad K-split budget");
  a->split_budget = workgroups;
  return SERL_OK;
}

int serl_agent_set_step(serl_agent* a, int64_t step) {
  SERL_REQUIRE(a && step >= 0, "bad argument");
  a->step = step;
  return SERL_OK;
}
int64_t serl_agent_get_step(serl_agent* a) { return a ? a->step : -1; }

int serl_agent_trunk_forward(serl_agent* a, const uint8_t* dev_frames, int n, float* dev_feats_out, void* stream) {
  SERL_REQUIRE(a && dev_frames && dev_feats_out, "NULL argument");
  SERL_REQUIRE(!a->state_only && !a->small, "this agent has no frozen trunk");
  SERL_HIP(hipSetDevice(a->cfg.device));
  return trunk_forward(a->tw, a->tws, dev_frames, n, dev_feats_out, (hipStream_t)stream, a->trunk_mode ? &a->tpk : nullptr);
}

static int check_batch(serl_agent* a, const serl_batch* b) {
  const serl_agent_cfg& c = a->cfg;
  SERL_REQUIRE(b && (b->frames || c.n_cam == 0) && b->state && b->action && b->reward && b->mask, "serl_batch has NULL members");
  SERL_REQUIRE(b->batch >= 1 && b->batch <= c.batch, "batch %d not in [1,%d]", b->batch, c.batch);
  SERL_REQUIRE(b->n_cam == c.n_cam && (c.n_cam == 0 || (b->H == c.H && b->W == c.W && b->C == 3)) &&
                   b->state_dim == c.state_dim && b->act_dim == c.act_dim, "serl_batch shape does not match the agent");
  return SERL_OK;
}

int serl_agent_encode_slot(serl_agent* a, const serl_batch* batch, int slot, void* stream) {
  return serl_agent_encode_slot_range(a, batch, slot, -1, kTrunkStages - 1, stream);
}

int serl_agent_encode_slot_range(serl_agent* a, const serl_batch* batch, int slot, int stage_begin, int stage_end, void* stream) {
  SERL_REQUIRE(a, "NULL agent");
  SERL_REQUIRE(slot >= 0 && slot < serl_agent::kSlots, "slot must be 0..%d", serl_agent::kSlots - 1);
  SERL_REQUIRE(stage_begin >= -1 && stage_begin <= stage_end && stage_end < kTrunkStages, "bad stage range [%d, %d]", stage_begin, stage_end);
  int rc = check_batch(a, batch);
  if (rc) return rc;
  SERL_HIP(hipSetDevice(a->cfg.device));
  hipStream_t st = (hipStream_t)stream;
  a->cur_slot[slot] = *batch;
  a->slot_valid[slot] = true;
  float* feats = a->feats_slot[slot];
  const serl_agent_cfg& c = a->cfg;
  const int B = batch->batch;
  const size_t fbytes = (size_t)c.H * c.W * 3;
  if (a->state_only || a->small) return SERL_OK;  // no frozen encoder: nothing to precompute
  if (B == c.batch) {  // frames [2][n_cam][B] are one contiguous run of 2*n_cam*B images
    // (sub-batching the run to keep activations in the Infinity Cache was measured: slower -- DESIGN.md)
    return trunk_forward(a->tw, a->tws, batch->frames, 2 * c.n_cam * B, feats, st, a->trunk_mode ? &a->tpk : nullptr, stage_begin, stage_end);
  }
  SERL_REQUIRE(stage_begin < 0 && stage_end == kTrunkStages - 1, "partial trunk passes need a full-size batch");
  for (int w = 0; w < 2; ++w)
    for (int k = 0; k < c.n_cam; ++k)
      RC(trunk_forward(a->tw, a->tws, batch->frames + ((size_t)(w * c.n_cam + k) * B) * fbytes, B,
                       feats + (((long)w * c.n_cam + k) * c.batch) * a->HW * 512, st, a->trunk_mode ? &a->tpk : nullptr));
  return SERL_OK;
}

int serl_agent_slot_features(serl_agent* a, int slot, float** dev_out, int64_t* count_out) {
  SERL_REQUIRE(a && dev_out && count_out, "NULL argument");
  SERL_REQUIRE(slot >= 0 && slot < serl_agent::kSlots, "slot must be 0..%d", serl_agent::kSlots - 1);
  SERL_REQUIRE(!a->state_only && !a->small, "this agent has no frozen-trunk features");
  *dev_out = a->feats_slot[slot];
  *count_out = 2LL * a->cfg.n_cam * a->cfg.batch * a->HW * 512;
  return SERL_OK;
}

int serl_agent_bind_slot(serl_agent* a, const serl_batch* batch, int slot) {
  SERL_REQUIRE(a, "NULL agent");
  SERL_REQUIRE(slot >= 0 && slot < serl_agent::kSlots, "slot must be 0..%d", serl_agent::kSlots - 1);
  int rc = check_batch(a, batch);
  if (rc) return rc;
  SERL_REQUIRE(batch->batch == a->cfg.batch, "externally computed features cover a full-size batch");
  a->cur_slot[slot] = *batch;
  a->slot_valid[slot] = true;
  return SERL_OK;
}

int serl_agent_select_slot(serl_agent* a, int slot) {
  SERL_REQUIRE(a, "NULL agent");
  SERL_REQUIRE(slot >= 0 && slot < serl_agent::kSlots && a->slot_valid[slot], "slot %d holds no encoded batch", slot);
  a->feats = a->feats_slot[slot];
  a->cur = a->cur_slot[slot];
  a->has_batch = true;
  return SERL_OK;
}

int serl_agent_encode(serl_agent* a, const serl_batch* batch, void* stream) {
  RC(serl_agent_encode_slot(a, batch, 0, stream));
  return serl_agent_select_slot(a, 0);
}

int serl_agent_begin_update(serl_agent* a, void* stream) {
  SERL_REQUIRE(a, "NULL agent");
  SERL_HIP(hipSetDevice(a->cfg.device));
  (void)stream;
  a->info_reset = true;  // the next critic step's info kernel zeroes the accumulators (no memset launch)
  return SERL_OK;
}

int serl_agent_set_shard(serl_agent* a, int64_t global_offset, int64_t global_batch) {
  SERL_REQUIRE(a, "NULL agent");
  SERL_REQUIRE(global_batch == 0 || (global_offset >= 0 && global_offset < global_batch), "bad shard [%lld of %lld]",
               (long long)global_offset, (long long)global_batch);
  a->shard_off = global_batch ? global_offset : 0;
  a->shard_global = global_batch;
  return SERL_OK;
}

int serl_agent_critic_grads(serl_agent* a, int off, int cnt, int global_count, const serl_noise* noise,
                            int redq_row, void* stream) {
  return serl_agent_critic_grads_bucketed(a, off, cnt, global_count, noise, redq_row, stream, nullptr);
}

int serl_agent_grad_bucket(serl_agent* a, int bucket, float** dev_ptr, int64_t* count) {
  SERL_REQUIRE(a && dev_ptr && count, "NULL argument");
  SERL_REQUIRE(bucket == 0 || bucket == 1, "bucket must be 0 or 1");
  const Offs& o = a->o;
  if (bucket == 0) { *dev_ptr = a->Gc + o.c_w1; *count = (o.Pc - o.c_w1) + kScalars; }
  else { *dev_ptr = a->Gc; *count = o.c_w1; }
  return SERL_OK;
}

int serl_agent_critic_grads_bucketed(serl_agent* a, int off, int cnt, int global_count, const serl_noise* noise,
                                     int redq_row, void* stream, void* event_bucket0) {
  SERL_REQUIRE(a && a->has_batch, "serl_agent_encode must run first");
  SERL_REQUIRE(off >= 0 && cnt >= 1 && off + cnt <= a->cur.batch && global_count >= cnt, "bad minibatch range");
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  hipStream_t st = (hipStream_t)stream;
  SERL_HIP(hipSetDevice(c.device));
  // REDQ subsample (sac.py:150-157): critic_subsample_size members drawn with replacement; cfg 0 = the launcher's 2,
  // -1 = None (the minimum runs over the whole ensemble and nothing is drawn)
  const int m_sub = c.critic_subsample_size == 0 ? 2 : (c.critic_subsample_size < 0 ? 0 : c.critic_subsample_size);
  SERL_REQUIRE(m_sub <= 16, "critic_subsample_size %d exceeds 16", m_sub);
  RedqSel sel{};
  sel.n = m_sub;
  if (noise && noise->redq_idx) {
    for (int k = 0; k < m_sub; ++k) sel.idx[k] = noise->redq_idx[m_sub * redq_row + k];
  } else {  // randint(0, ensemble) with replacement
    uint64_t r = c.seed * 0x9E3779B97F4A7C15ull + (++a->noise_ctr) * 0xBF58476D1CE4E5B9ull;
    for (int k = 0; k < m_sub; ++k) {
      r ^= r >> 29; r *= 0x94D049BB133111EBull; r ^= r >> 32;
      sel.idx[k] = (int)(r % c.ensemble);
      r += 0x9E3779B97F4A7C15ull;
    }
  }
  for (int k = 0; k < m_sub; ++k) SERL_REQUIRE(sel.idx[k] >= 0 && sel.idx[k] < c.ensemble, "REDQ index out of range");
  a->pg_ncs = a->pg_nwg = 0;
  a->pg_defer = true;
  const int A = c.act_dim;
  if (a->fuse) {
    const FusedNoise fz = fetch_noise_fused(a, noise ? noise->eps_next : nullptr, noise ? noise->mask_next : nullptr, 0,
                                            noise && noise->key_eps_next ? noise->key_eps_next + 2 * redq_row : nullptr,
                                            noise && noise->key_mask_next ? noise->key_mask_next + 2 * c.n_cam * redq_row : nullptr);
    const long tf_row0 = tf_row0_of(a, cnt);
    EncJob ej[3] = {{a->theta, 1, fz.mask, &a->encP, nullptr, nullptr},
                    {a->theta_t, 1, nullptr, &a->encT, nullptr, nullptr},
                    {a->theta, 0, nullptr, &a->encO, a->cur.action + (long)off * A, a->crit.x + a->E}};
    ej[0].gen_mask = fz.gen_mask; ej[0].mask_seed = fz.mask_seed;
    ej[0].tf_key = fz.tf_mask; ej[0].tf_rows = global_count; ej[0].tf_row0 = tf_row0;
    RC(encode_multi(a, ej, 3, off, cnt, st));
    PolJob pj{a->theta, &a->pol, a->encP.enc, a->encP.ld, fz.eps ? fz.eps + (long)off * A : nullptr, a->critT.x + a->E, a->XA, nullptr,
              c.backup_entropy ? a->aux + X_ALPHA : nullptr};
    pj.eps_out = fz.eps_out + (long)off * A; pj.eps_seed = fz.eps_seed; pj.eps_row0 = a->shard_off + off;
    pj.tf_key = fz.tf_eps; pj.tf_rows = global_count; pj.tf_row0 = tf_row0;
    RC(policy_fwd_multi(a, &pj, 1, cnt, st));
    const CritJob cj[2] = {{a->theta_t, &a->critT}, {a->theta, &a->crit}};
    RC(critic_fwd_multi(a, cj, 2, cnt, st));
    // REDQ target + loss: a rider workgroup of the LayerNorm-backward launch that consumes dQ (no critic_loss launch)
    const LossArgs L{1, a->critT.q, a->crit.q, a->cur.reward + off, a->cur.mask + off, sel, c.ensemble, cnt, c.discount,
                     1.0f / ((float)c.ensemble * (float)global_count), a->ytgt, a->dq, a->SC, a->Gc + o.c_hb, a->state_only ? 1 : 0,
                     c.backup_entropy ? a->pol.logp : nullptr, c.backup_entropy ? a->aux + X_ALPHA : nullptr};
    SERL_REQUIRE(!a->state_only || c.ensemble <= 16, "per-member head bias supports ensembles of at most 16");
    RC(critic_bwd(a, a->theta, a->crit, cnt, true, st, a->dq, 0.f, &L));
    if (!a->state_only) RC(enc_proprio_bwd_critic_fused(a, a->theta, a->encO, off, cnt, st, event_bucket0));
    else if (event_bucket0) {
      RC(flush_param_grads(a, st));
      SERL_HIP(hipEventRecord((hipEvent_t)event_bucket0, st));
    }
    RC(flush_param_grads(a, st));
    a->last_global = global_count;
    return SERL_OK;
  }
  const float* eps; const uint8_t* mask;
  NoiseBatch nb;
  const float* g_eps = noise ? noise->eps_next : nullptr;
  const uint8_t* g_mask = noise ? noise->mask_next : nullptr;
  RC(jax_noise_tensors(a, noise && noise->key_eps_next ? noise->key_eps_next + 2 * redq_row : nullptr,
                       noise && noise->key_mask_next ? noise->key_mask_next + 2 * c.n_cam * redq_row : nullptr, 0, off, cnt, global_count,
                       st, &g_eps, &g_mask));
  fetch_noise(a, nb, g_eps, g_mask, 0, a->cur.batch, &eps, &mask);
  RC(nb.flush(st));
  // the three encoder passes of the critic loss in one set of launches: online policy input at next_obs
  // (dropout), target-critic input at next_obs (target_params, train=False), online-critic input at obs
  // (train=False; the batch's actions ride along into [enc | action])
  const EncJob ej[3] = {{a->theta, 1, mask, &a->encP, nullptr, nullptr},
                        {a->theta_t, 1, nullptr, &a->encT, nullptr, nullptr},
                        {a->theta, 0, nullptr, &a->encO, a->cur.action + (long)off * A, a->crit.x + a->E}};
  RC(encode_multi(a, ej, 3, off, cnt, st));
  // backup_entropy (sac.py:174-176) needs alpha = softplus(lagrange) of the online parameters next to the per-sample log-probs
  const PolJob pj{a->theta, &a->pol, a->encP.enc, a->encP.ld, eps + (long)off * A, a->critT.x + a->E, a->XA, nullptr,
                  c.backup_entropy ? a->aux + X_ALPHA : nullptr};
  RC(policy_fwd_multi(a, &pj, 1, cnt, st));
  const CritJob cj[2] = {{a->theta_t, &a->critT}, {a->theta, &a->crit}};  // target and online ensembles together
  RC(critic_fwd_multi(a, cj, 2, cnt, st));
  const float inv_norm = 1.0f / ((float)c.ensemble * (float)global_count);
  RC(critic_loss(a->critT.q, a->crit.q, a->cur.reward + off, a->cur.mask + off, sel, c.ensemble, cnt, c.discount,
                 inv_norm, a->ytgt, a->dq, a->SC, a->Gc + o.c_hb, st, a->state_only,
                 c.backup_entropy ? a->pol.logp : nullptr, c.backup_entropy ? a->aux + X_ALPHA : nullptr));
  RC(critic_bwd(a, a->theta, a->crit, cnt, true, st, a->dq, 0.f));
  if (!a->state_only)
    RC(proprio_bwd(a, a->theta, a->dx + (long)c.n_cam * c.bottleneck, a->XA, a->crit.x + (long)c.n_cam * c.bottleneck,
                   a->XA, a->encO, 0, off, cnt, a->Gc, 0, st));
  if (event_bucket0) {   // bucket 0 (ensemble | head | proprio | scalars) is final here: publish it before the encoder heads
    RC(flush_param_grads(a, st));
    SERL_HIP(hipEventRecord((hipEvent_t)event_bucket0, st));
  }
  if (!a->state_only) RC(encode_bwd_critic(a, a->theta, a->encO, off, cnt, st));
  RC(flush_param_grads(a, st));
  a->last_global = global_count;
  return SERL_OK;
}

int serl_agent_actor_grads(serl_agent* a, int global_count, const serl_noise* noise, void* stream) {
  SERL_REQUIRE(a && a->has_batch, "serl_agent_encode must run first");
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  hipStream_t st = (hipStream_t)stream;
  SERL_HIP(hipSetDevice(c.device));
  const int cnt = a->cur.batch, A = c.act_dim, Hd = c.hidden;
  SERL_REQUIRE(global_count >= cnt, "global_count < local batch");
  a->pg_ncs = a->pg_nwg = 0;
  a->pg_defer = true;
  const float* eps_pi; const uint8_t* mask_pi; const float* eps_t; const uint8_t* mask_t;
  hipStream_t s0 = st;
  if (a->fuse) {
    const FusedNoise fp = fetch_noise_fused(a, noise ? noise->eps_pi : nullptr, noise ? noise->mask_obs_pi : nullptr, 1,
                                            noise ? noise->key_eps_pi : nullptr, noise ? noise->key_mask_obs_pi : nullptr);
    const FusedNoise ft = fetch_noise_fused(a, noise ? noise->eps_temp : nullptr, noise ? noise->mask_next_temp : nullptr, 2,
                                            noise ? noise->key_eps_temp : nullptr, noise ? noise->key_mask_next_temp : nullptr);
    const long tf_row0 = tf_row0_of(a, cnt);
    EncJob ej[3] = {{a->theta, 1, ft.mask, &a->encT, nullptr, nullptr},
                    {a->theta, 0, nullptr, &a->encO, nullptr, nullptr},
                    {a->theta, 0, fp.mask, &a->encP, nullptr, nullptr}};
    ej[0].gen_mask = ft.gen_mask; ej[0].mask_seed = ft.mask_seed;
    ej[2].gen_mask = fp.gen_mask; ej[2].mask_seed = fp.mask_seed;
    ej[0].tf_key = ft.tf_mask; ej[0].tf_rows = global_count; ej[0].tf_row0 = tf_row0;
    ej[2].tf_key = fp.tf_mask; ej[2].tf_rows = global_count; ej[2].tf_row0 = tf_row0;
    RC(encode_multi(a, ej, 3, 0, cnt, s0));
    PolJob pj[2] = {{a->theta, &a->polT, a->encT.enc, a->encT.ld, ft.eps, a->act_tmp, A, a->SC + S_LOGP_NEXT, a->aux + X_ALPHA},
                    {a->theta, &a->pol, a->encP.enc, a->encP.ld, fp.eps, a->crit.x + a->E, a->XA, a->SC + S_LOGP, nullptr}};
    pj[0].eps_out = ft.eps_out; pj[0].eps_seed = ft.eps_seed; pj[0].eps_row0 = a->shard_off;
    pj[1].eps_out = fp.eps_out; pj[1].eps_seed = fp.eps_seed; pj[1].eps_row0 = a->shard_off;
    pj[0].tf_key = ft.tf_eps; pj[0].tf_rows = global_count; pj[0].tf_row0 = tf_row0;
    pj[1].tf_key = fp.tf_eps; pj[1].tf_rows = global_count; pj[1].tf_row0 = tf_row0;
    RC(policy_fwd_multi(a, pj, 2, cnt, s0));
    eps_pi = fp.eps ? fp.eps : fp.eps_out;   // (the backward reads the draws the head epilogue used)
    const CritJob cj{a->theta, &a->crit};
    RC(critic_fwd_multi(a, &cj, 1, cnt, s0));
    RC(critic_bwd(a, a->theta, a->crit, cnt, false, s0, nullptr, -1.0f / ((float)c.ensemble * (float)global_count)));
    RC(policy_dist_bwd(a->dx + a->E, a->XA, a->crit.x + a->E, a->XA, a->pol.pre, a->pol.std, eps_pi, a->aux + X_ALPHA,
                       1.0f / (float)global_count, cnt, A, c.std_min, c.std_max, a->dpre, a->crit.q, c.ensemble,
                       a->SC + S_QPI, s0));
    const long hs = o.a_Ws - o.a_Wm;
    float* Ga = a->Ga;
    const long b0 = o.Pa0;
    RC(wgrad(a, a->pol.m.h2, Hd, 0, a->dpre, A, (long)cnt * A, Ga + (o.a_Wm - b0), A, hs, 2, Hd, A, cnt, s0));
    SERL_REQUIRE(a->pg_ncs < kMaxColsum, "no room for the head-bias gradient job");
    a->pg_cs[a->pg_ncs++] = Colsum3Args{a->dpre, nullptr, nullptr, 2, cnt, A, nullptr, Ga + (o.a_bm - b0), nullptr, hs, 1};
    // dh2 = dmean W_mean^T + dlog_std W_logstd^T: both products and their sum in one launch
    RC(igrad_sum(a, a->dpre, A, (long)cnt * A, a->theta + o.a_Wm, A, hs, a->dh2, Hd, 2, cnt, Hd, A, s0));
    RC(dense_ln_tanh_bwd(a, a->dh2, Hd, 0, a->pol.m.h2, Hd, 0, a->pol.m.xh2, a->pol.m.rs2, a->theta + o.a_g2, 0, 1, cnt, Hd,
                         a->da2, a->dg2, Ga, o.a_g2 - b0, o.a_be2 - b0, o.a_b2 - b0, 0, s0));
    RC(wgrad(a, a->pol.m.h1, Hd, 0, a->da2, Hd, 0, Ga + (o.a_w2 - b0), Hd, 0, 1, Hd, Hd, cnt, s0));
    RC(igrad(a->da2, Hd, 0, a->theta + o.a_w2, Hd, 0, a->dh1, Hd, 0, 1, cnt, Hd, Hd, s0));
    RC(dense_ln_tanh_bwd(a, a->dh1, Hd, 0, a->pol.m.h1, Hd, 0, a->pol.m.xh1, a->pol.m.rs1, a->theta + o.a_g1, 0, 1, cnt, Hd,
                         a->da1, a->dg1, Ga, o.a_g1 - b0, o.a_be1 - b0, o.a_b1 - b0, 0, s0));
    RC(wgrad(a, a->encP.enc, a->encP.ld, 0, a->da1, Hd, 0, Ga + (o.a_w1 - b0), Hd, 0, 1, a->E, Hd, cnt, s0));
    if (!a->state_only) {
      const long pc = (long)c.n_cam * c.bottleneck;
      RC(igrad(a->da1, Hd, 0, a->theta + o.a_w1 + pc * Hd, Hd, 0, a->dprop_y, c.proprio_dim, 0, 1, cnt, c.proprio_dim, Hd, s0));
      RC(proprio_bwd(a, a->theta, a->dprop_y, c.proprio_dim, a->encP.enc + pc, a->encP.ld, a->encP, 0, 0, cnt, Ga, b0, s0));
    }
    RC(flush_param_grads(a, s0));
    a->last_global = global_count;
    return SERL_OK;
  }
  NoiseBatch nb;
  const float* g_eps_pi = noise ? noise->eps_pi : nullptr; const uint8_t* g_mask_pi = noise ? noise->mask_obs_pi : nullptr;
  const float* g_eps_t = noise ? noise->eps_temp : nullptr; const uint8_t* g_mask_t = noise ? noise->mask_next_temp : nullptr;
  RC(jax_noise_tensors(a, noise ? noise->key_eps_pi : nullptr, noise ? noise->key_mask_obs_pi : nullptr, 1, 0, cnt, global_count, st,
                       &g_eps_pi, &g_mask_pi));
  RC(jax_noise_tensors(a, noise ? noise->key_eps_temp : nullptr, noise ? noise->key_mask_next_temp : nullptr, 2, 0, cnt, global_count, st,
                       &g_eps_t, &g_mask_t));
  fetch_noise(a, nb, g_eps_pi, g_mask_pi, 1, cnt, &eps_pi, &mask_pi);
  fetch_noise(a, nb, g_eps_t, g_mask_t, 2, cnt, &eps_t, &mask_t);
  RC(nb.flush(st));
  // encoder passes of the actor step in one set of launches: temperature loss input (next_obs, dropout;
  // sac.py:223-234), critic-side encoding of obs (train=False), policy input at obs (dropout; sac.py:193-221)
  const EncJob ej[3] = {{a->theta, 1, mask_t, &a->encT, nullptr, nullptr},
                        {a->theta, 0, nullptr, &a->encO, nullptr, nullptr},
                        {a->theta, 0, mask_pi, &a->encP, nullptr, nullptr}};
  RC(encode_multi(a, ej, 3, 0, cnt, s0));
  const PolJob pj[2] = {{a->theta, &a->polT, a->encT.enc, a->encT.ld, eps_t, a->act_tmp, A, a->SC + S_LOGP_NEXT, a->aux + X_ALPHA},
                        {a->theta, &a->pol, a->encP.enc, a->encP.ld, eps_pi, a->crit.x + a->E, a->XA, a->SC + S_LOGP, nullptr}};
  RC(policy_fwd_multi(a, pj, 2, cnt, s0));
  const CritJob cj{a->theta, &a->crit};
  RC(critic_fwd_multi(a, &cj, 1, cnt, s0));
  RC(critic_bwd(a, a->theta, a->crit, cnt, false, s0, nullptr, -1.0f / ((float)c.ensemble * (float)global_count)));
  RC(policy_dist_bwd(a->dx + a->E, a->XA, a->crit.x + a->E, a->XA, a->pol.pre, a->pol.std, eps_pi, a->aux + X_ALPHA,
                     1.0f / (float)global_count, cnt, A, c.std_min, c.std_max, a->dpre, a->crit.q, c.ensemble,
                     a->SC + S_QPI, s0));
  // policy heads (mean, log_std): nbatch = 2 with uniform stride; parameter gradients off the critical chain
  const long hs = o.a_Ws - o.a_Wm;
  float* Ga = a->Ga;
  const long b0 = o.Pa0;
  RC(wgrad(a, a->pol.m.h2, Hd, 0, a->dpre, A, (long)cnt * A, Ga + (o.a_Wm - b0), A, hs, 2, Hd, A, cnt, s0));
  RC(colsum(a->dpre, nullptr, 2, cnt, A, Ga + (o.a_bm - b0), hs, false, s0));
  RC(igrad(a->dpre, A, (long)cnt * A, a->theta + o.a_Wm, A, hs, a->slabs, Hd, (long)cnt * Hd, 2, cnt, Hd, A, s0));
  RC(reduce_slabs(a->slabs, 2, (long)cnt * Hd, 1, cnt, Hd, nullptr, 0, a->dh2, Hd, 0, false, s0));
  RC(dense_ln_tanh_bwd(a, a->dh2, Hd, 0, a->pol.m.h2, Hd, 0, a->pol.m.xh2, a->pol.m.rs2, a->theta + o.a_g2, 0, 1, cnt, Hd,
                       a->da2, a->dg2, Ga, o.a_g2 - b0, o.a_be2 - b0, o.a_b2 - b0, 0, s0));
  RC(wgrad(a, a->pol.m.h1, Hd, 0, a->da2, Hd, 0, Ga + (o.a_w2 - b0), Hd, 0, 1, Hd, Hd, cnt, s0));
  RC(igrad(a->da2, Hd, 0, a->theta + o.a_w2, Hd, 0, a->dh1, Hd, 0, 1, cnt, Hd, Hd, s0));
  RC(dense_ln_tanh_bwd(a, a->dh1, Hd, 0, a->pol.m.h1, Hd, 0, a->pol.m.xh1, a->pol.m.rs1, a->theta + o.a_g1, 0, 1, cnt, Hd,
                       a->da1, a->dg1, Ga, o.a_g1 - b0, o.a_be1 - b0, o.a_b1 - b0, 0, s0));
  RC(wgrad(a, a->encP.enc, a->encP.ld, 0, a->da1, Hd, 0, Ga + (o.a_w1 - b0), Hd, 0, 1, a->E, Hd, cnt, s0));
  // image codes are stop-gradiented (encoding.py:48-49); only the proprio slice of d_enc is needed
  if (!a->state_only) {
    const long pc = (long)c.n_cam * c.bottleneck;
    RC(igrad(a->da1, Hd, 0, a->theta + o.a_w1 + pc * Hd, Hd, 0, a->dprop_y, c.proprio_dim, 0, 1, cnt, c.proprio_dim, Hd, s0));
    RC(proprio_bwd(a, a->theta, a->dprop_y, c.proprio_dim, a->encP.enc + pc, a->encP.ld, a->encP, 0, 0, cnt, Ga, b0, s0));
  }
  RC(flush_param_grads(a, s0));
  a->last_global = global_count;
  return SERL_OK;
}

int serl_agent_apply(serl_agent* a, int which, float info_weight, void* stream) {
  SERL_REQUIRE(a, "NULL agent");
  SERL_REQUIRE(which >= 1 && which <= 7, "bad `which` (a non-empty set of SERL_NET_* bits)");
  const serl_agent_cfg& c = a->cfg;
  const Offs& o = a->o;
  hipStream_t st = (hipStream_t)stream;
  SERL_HIP(hipSetDevice(c.device));
  const bool crit = which & SERL_NET_CRITIC, act = which & SERL_NET_ACTOR, temp = which & SERL_NET_TEMPERATURE;
  AdamArgs ad{};
  ad.theta = a->theta; ad.theta_target = a->theta_t;
  ad.P = o.P; ad.Pc = o.Pc; ad.Pa0 = o.Pa0; ad.Pa1 = o.Pa1;
  ad.g_critic = a->Gc; ad.g_actor = a->Ga;
  ad.m_c = a->m_c; ad.v_c = a->v_c; ad.m_a = a->m_a; ad.v_a = a->v_a; ad.m_t = a->m_t; ad.v_t = a->v_t;
  ad.sum_logp_next = a->SC + S_LOGP_NEXT; ad.temp_grad_out = a->aux + X_TGRAD;
  ad.critic_on = crit ? 1 : 0; ad.actor_on = act ? 1 : 0; ad.temp_on = temp ? 1 : 0;
  ad.ema_on = crit ? 1 : 0;   // sac.py:284-285
  // all three txs share count == step (common.py:136-168 steps every optimizer on every update)
  ad.lr_a = lr_at(c, a->step, SERL_TX_ACTOR);
  ad.lr_c = lr_at(c, a->step, SERL_TX_CRITIC);
  ad.lr_t = lr_at(c, a->step, SERL_TX_TEMPERATURE);
  const int64_t t = a->step + 1;
  ad.bc1 = 1.0f - powf(0.9f, (float)t);
  ad.bc2 = 1.0f - powf(0.999f, (float)t);
  ad.tau = c.tau; ad.target_entropy = c.target_entropy;
  ad.inv_batch = a->last_global > 0 ? 1.0f / (float)a->last_global : 0.f;
  ad.frozen = a->trunk; ad.frozen_target = a->trunk_t; ad.n_frozen = a->trunk_count;  // common.py:124-134 covers every leaf
  const bool any_wd = c.tx_weight_decay_on[SERL_TX_ACTOR] || c.tx_weight_decay_on[SERL_TX_CRITIC] || c.tx_weight_decay_on[SERL_TX_TEMPERATURE];
  if (!any_wd && a->trunk_count > 0) {   // frozen leaves only ever see the EMA: count the step, apply it lazily (flush_trunk_ema)
    ad.n_frozen = 0;
    if (crit) a->trunk_ema_pending += 1;
  }
  ad.wd_a = c.tx_weight_decay_on[SERL_TX_ACTOR] ? c.tx_weight_decay[SERL_TX_ACTOR] : 0.f;
  ad.wd_c = c.tx_weight_decay_on[SERL_TX_CRITIC] ? c.tx_weight_decay[SERL_TX_CRITIC] : 0.f;
  ad.wd_t = c.tx_weight_decay_on[SERL_TX_TEMPERATURE] ? c.tx_weight_decay[SERL_TX_TEMPERATURE] : 0.f;
  ad.clip_a = c.tx_clip_norm[SERL_TX_ACTOR]; ad.clip_c = c.tx_clip_norm[SERL_TX_CRITIC]; ad.clip_t = c.tx_clip_norm[SERL_TX_TEMPERATURE];
  ad.norm2 = a->aux + X_NORM2;
  if ((ad.clip_c > 0.f && crit) || (ad.clip_a > 0.f && act))
    RC(grad_norm2(a->Gc, o.Pc, a->Ga, o.Pa1 - o.Pa0, a->aux + X_NORM2, st));
  ad.info_mode = (crit ? 1 : 0) | ((act || temp) ? 2 : 0);
  ad.info_reset = (crit && a->info_reset) ? 1 : 0;
  if (crit) a->info_reset = false;
  ad.scalars = a->SC; ad.alpha = a->aux + X_ALPHA; ad.info_acc = a->info_acc;
  ad.info_w = info_weight;
  ad.inv_eb = a->last_global > 0 ? 1.0f / ((float)c.ensemble * (float)a->last_global) : 0.f;
  RC(adam_ema(ad, st));
  a->step += 1;
  a->lr_last[SERL_TX_ACTOR] = ad.lr_a; a->lr_last[SERL_TX_CRITIC] = ad.lr_c; a->lr_last[SERL_TX_TEMPERATURE] = ad.lr_t;
  return SERL_OK;
}

int serl_agent_grad_view(serl_agent* a, int which, float** dev_ptr, int64_t* count) {
  SERL_REQUIRE(a && dev_ptr && count, "NULL argument");
  SERL_REQUIRE(which >= 1 && which <= 7, "bad `which`");
  const Offs& o = a->o;
  const bool crit = which & SERL_NET_CRITIC, at = which & (SERL_NET_ACTOR | SERL_NET_TEMPERATURE);
  if (crit && at) { *dev_ptr = a->G; *count = o.Pc + kScalars + (o.Pa1 - o.Pa0); }
  else if (crit) { *dev_ptr = a->Gc; *count = o.Pc + kScalars; }
  else { *dev_ptr = a->SC; *count = kScalars + (o.Pa1 - o.Pa0); }
  return SERL_OK;
}

int serl_agent_update(serl_agent* a, const serl_batch* batch, int nets, const serl_noise* noise, void* stream) {
  SERL_REQUIRE(a && batch, "NULL argument");
  SERL_REQUIRE(nets >= 1 && nets <= 7, "Invalid gradient steps: %d", nets);   // sac.py:272-274
  RC(serl_agent_begin_update(a, stream));
  RC(serl_agent_encode(a, batch, stream));
  if (nets & SERL_NET_CRITIC) RC(serl_agent_critic_grads(a, 0, batch->batch, batch->batch, noise, 0, stream));
  if (nets & (SERL_NET_ACTOR | SERL_NET_TEMPERATURE)) RC(serl_agent_actor_grads(a, batch->batch, noise, stream));
  return serl_agent_apply(a, nets, 1.0f, stream);
}

int serl_agent_update_critics(serl_agent* a, const serl_batch* batch, const serl_noise* noise, void* stream) {
  RC(serl_agent_begin_update(a, stream));
  RC(serl_agent_encode(a, batch, stream));
  RC(serl_agent_critic_grads(a, 0, batch->batch, batch->batch, noise, 0, stream));
  return serl_agent_apply(a, SERL_APPLY_CRITIC, 1.0f, stream);
}

int serl_agent_update_high_utd(serl_agent* a, const serl_batch* batch, int utd_ratio, const serl_noise* noise,
                               void* stream) {
  SERL_REQUIRE(a && batch, "NULL argument");
  SERL_REQUIRE(utd_ratio >= 1 && batch->batch % utd_ratio == 0,
               "Batch size %d must be divisible by UTD ratio %d", batch->batch, utd_ratio);  // sac.py:561-563
  RC(serl_agent_begin_update(a, stream));
  RC(serl_agent_encode(a, batch, stream));
  const int mb = batch->batch / utd_ratio;
  for (int i = 0; i < utd_ratio; ++i) {
    RC(serl_agent_critic_grads(a, i * mb, mb, mb, noise, i, stream));
    RC(serl_agent_apply(a, SERL_APPLY_CRITIC, 1.0f / (float)utd_ratio, stream));
  }
  RC(serl_agent_actor_grads(a, batch->batch, noise, stream));
  return serl_agent_apply(a, SERL_APPLY_ACTOR_TEMP, 1.0f, stream);
}

int serl_agent_read_info(serl_agent* a, serl_info* out, void* stream) {
  SERL_REQUIRE(a && out, "NULL argument");
  SERL_HIP(hipSetDevice(a->cfg.device));
  float acc[I_N];
  SERL_HIP(hipMemcpyAsync(acc, a->info_acc, sizeof(acc), hipMemcpyDeviceToHost, (hipStream_t)stream));
  SERL_HIP(hipStreamSynchronize((hipStream_t)stream));
  out->critic_loss = acc[I_CL]; out->predicted_qs = acc[I_PQ]; out->target_qs = acc[I_TQ];
  out->actor_loss = acc[I_AL]; out->temperature = acc[I_TEMP]; out->entropy = acc[I_ENT];
  out->temperature_loss = acc[I_TL];
  out->actor_lr = a->lr_last[SERL_TX_ACTOR]; out->critic_lr = a->lr_last[SERL_TX_CRITIC];
  out->temperature_lr = a->lr_last[SERL_TX_TEMPERATURE];
  return SERL_OK;
}

int serl_agent_sample_actions(serl_agent* a, const uint8_t* dev_frames, const float* dev_state, int n,
                              const float* dev_eps, float* dev_out_actions, void* stream) {
  SERL_REQUIRE(a && dev_state && dev_out_actions, "NULL argument");
  const serl_agent_cfg& c = a->cfg;
  SERL_REQUIRE(dev_frames || c.n_cam == 0, "NULL frames");
  SERL_REQUIRE(n >= 1 && n <= c.batch, "n %d not in [1,%d]", n, c.batch);
  hipStream_t st = (hipStream_t)stream;
  SERL_HIP(hipSetDevice(c.device));
  // trunk on [n_cam][n] images -> feats slot 0; state goes through a temporary serl_batch view
  const size_t fbytes = (size_t)c.H * c.W * 3;
  for (int k = 0; k < c.n_cam && !a->small; ++k)
    RC(trunk_forward(a->tw, a->tws, dev_frames + (size_t)k * n * fbytes, n, a->feats_slot[serl_agent::kSlots] + ((long)k * c.batch) * a->HW * 512, st, a->trunk_mode ? &a->tpk : nullptr));
  serl_batch saved = a->cur;
  const bool had = a->has_batch;
  float* saved_feats = a->feats;
  a->feats = a->feats_slot[serl_agent::kSlots];
  a->cur = serl_batch{};
  a->cur.batch = n;
  a->cur.state = const_cast<float*>(dev_state);
  a->cur.frames = const_cast<uint8_t*>(dev_frames);   // (the SmallEncoder reads the pixels in encode_multi)
  const EncJob ej{a->theta, 0, nullptr, &a->encP, nullptr, nullptr};  // train=False: no dropout
  RC(encode_multi(a, &ej, 1, 0, n, st));
  if (!dev_eps) {  // argmax -> mode = tanh(mean): zero noise
    RC(fill(a->eps_buf[0], 0.f, (long)n * c.act_dim, st));
    dev_eps = a->eps_buf[0];
  }
  PolJob pj{a->theta, &a->pol, a->encP.enc, a->encP.ld, dev_eps, dev_out_actions, c.act_dim, nullptr, nullptr};
  pj.eps_out = a->eps_buf[0];
  RC(policy_fwd_multi(a, &pj, 1, n, st));
  a->cur = saved;
  a->has_batch = had;
  a->feats = saved_feats;
  return SERL_OK;
}

int64_t serl_debug_chain_launches(void) { return (int64_t)g_chain_launches; }

int serl_agent_trunk_plan(serl_agent* a, char* out, int cap) {
  SERL_REQUIRE(a && out && cap > 0, "bad argument");
  const TrunkPlan& p = a->tws.plan;
  std::string s = "images=" + std::to_string(p.images) + " pool=" + std::to_string(p.pool) + " raw_b0=" + std::to_string(p.raw_b0);
  static const char* kNames[3] = {"conv0", "conv1", "proj"};
  for (int i = 0; i < kTrunkStages; ++i)
    for (int k = 0; k < 3; ++k) {
      const TrunkPlan::L& l = p.conv[i][k];
      if (!l.kern) continue;
      s += " b" + std::to_string(i) + "_" + kNames[k] + "=" + std::string(1, l.kern) + "/" + std::to_string(l.cfg) + "/" +
           std::to_string(l.pmode) + "/f" + std::to_string(l.fused) + (l.ksplit > 1 ? "k" + std::to_string(l.ksplit) : std::string());
    }
  snprintf(out, cap, "%s", s.c_str());
  return SERL_OK;
}

int serl_agent_debug_get(serl_agent* a, const char* what, float* host_out, int64_t count) {
  SERL_REQUIRE(a && what && host_out, "NULL argument");
  SERL_HIP(hipSetDevice(a->cfg.device));
  const Offs& o = a->o;
  const std::string w(what);
  const float* p = nullptr;
  long n = 0;
  const serl_agent_cfg& c = a->cfg;
  if (w == "ctr_nonzero") {
    // The last-arriver epilogues (heads.hip) rest on ONE invariant across launches: every arrival counter is zero when a launch
    // starts (the workgroup that draws the last ticket re-zeroes it with a memory-side store).  A lost arrival, a double ticket or a
    // late re-zero leaves a counter non-zero for the NEXT launch, which then never (or twice) reduces a tile.  host_out[0] = the
    // number of non-zero counters after a device synchronise: 0 at every quiescent point (the stress tests assert it).
    SERL_REQUIRE(count == 1, "tap 'ctr_nonzero' holds one value");
    SERL_HIP(hipDeviceSynchronize());
    std::vector<int> h((size_t)kCtrPerLane * kCtrLanes);
    SERL_HIP(hipMemcpy(h.data(), a->ctr, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
    long nz = 0;
    for (int v : h) nz += v != 0;
    host_out[0] = (float)nz;
    return SERL_OK;
  }
  if (w == "g_critic") { p = a->Gc; n = o.Pc; }
  else if (w == "g_actor") { p = a->Ga; n = o.Pa1 - o.Pa0; }
  else if (w == "scalars") { p = a->SC; n = kScalars; }
  else if (w == "q") { p = a->crit.q; n = (long)c.ensemble * c.batch; }
  else if (w == "q_target") { p = a->critT.q; n = (long)c.ensemble * c.batch; }
  else if (w == "target_q") { p = a->ytgt; n = c.batch; }
  else if (w == "x") { p = a->crit.x; n = (long)c.batch * a->XA; }
  else if (w == "x_target") { p = a->critT.x; n = (long)c.batch * a->XA; }
  else if (w == "feats") { p = a->feats; n = 2L * c.n_cam * c.batch * a->HW * 512; }
  else if (w == "logp") { p = a->pol.logp; n = c.batch; }
  else if (w == "dx") { p = a->dx; n = (long)c.batch * a->XA; }
  else { set_error("unknown debug tap '%s'", what); return SERL_ERR_INVALID; }
  SERL_REQUIRE(count <= n && count > 0, "tap '%s' holds %ld floats, asked for %lld", what, n, (long long)count);
  SERL_HIP(hipDeviceSynchronize());
  SERL_HIP(hipMemcpy(host_out, p, sizeof(float) * count, hipMemcpyDeviceToHost));
  return SERL_OK;
}

int serl_agent_debug_set(serl_agent* a, const char* what, const float* host, int64_t count) {
  SERL_REQUIRE(a && what && host, "NULL argument");
  SERL_HIP(hipSetDevice(a->cfg.device));
  const Offs& o = a->o;
  const std::string w(what);
  float* p = nullptr;
  long n = 0;
  if (w == "g_critic") { p = a->Gc; n = o.Pc; }
  else if (w == "g_actor") { p = a->Ga; n = o.Pa1 - o.Pa0; }
  else if (w == "scalars") { p = a->SC; n = kScalars; }
  else { set_error("unknown debug tap '%s'", what); return SERL_ERR_INVALID; }
  SERL_REQUIRE(count <= n && count > 0, "tap '%s' holds %ld floats, got %lld", what, n, (long long)count);
  SERL_HIP(hipDeviceSynchronize());
  SERL_HIP(hipMemcpy(p, host, sizeof(float) * count, hipMemcpyHostToDevice));
  if (a->last_global <= 0) a->last_global = 1;
  return SERL_OK;
}

}  // extern "C"
