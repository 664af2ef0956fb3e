// Host-callable launchers of heads.hip (internal, not part of the C ABI).
#pragma once
#include "internal.h"

namespace serl {

extern long g_chain_launches;   // kernels launched by heads.hip since the library was loaded (diagnostic)

// The update chain is a long sequence of small dependent kernels, each costing ~5 us of launch/drain latency
// whatever its size: independent instances of the same kind of work (the three EncodingWrapper passes of a
// loss, the online and target critics, the two policy evaluations of the actor step) share one launch.
constexpr int kMaxMulti = 4;
template <typename T>
struct Multi { T v[kMaxMulti]; };
constexpr int kMaxGemmGroups = 6;
int gemm_f32_multi(const GemmDesc* groups, int n, hipStream_t stream);  // same operand layout in every group

int reduce_slabs(const float* slabs, int S, long slab_stride, int groups, int rows, int N, const float* bias,
                 long bias_gstride, float* out, long ld_out, long out_gstride, bool accumulate,
                 hipStream_t stream, float scale = 1.0f);

int ln_tanh_fwd_multi(const LnFwdArgs* a, int n, int D, hipStream_t stream);

// REDQ target + critic loss as a rider workgroup of the LayerNorm-backward launch that consumes dQ (sac.py:142-191)
struct RedqSel { int n; int idx[16]; };
struct LossArgs {
  int on;
  const float *qt, *q, *reward, *mask; RedqSel sel; int E, B; float discount, inv_norm;
  float *y_out, *dq, *scalars, *dbias; int per_member; const float *logp_next, *alpha;
};
struct LnBwdArgs {
  const float* dy; long ld_dy; long dy_goff;  // same addressing as LnFwdArgs::y (ignored in rank-1 mode)
  // rank-1 mode (gradient through the critic head): dy[row][col] = (dq ? dq[row] : dq_const) * dq_w[col]
  const float* dq; const float* dq_w; float dq_const;
  long dq_w_gstride;  // 0: shared head vector; else per-group head vectors
  const float* y; long ld_y; long y_goff;
  const float* xhat; const float* rstd;
  const float* gamma; long pstride;
  int rows, rows_per_group;
  float* dx;  // [rows][D] gradient wrt the pre-activation
  float* dg;  // [rows][D] dy*(1-y^2)  (-> dbeta = colsum(dg), dgamma = colsum(dg*xhat))
  int D;          // ln_tanh_bwd_multi only: 64 or 256
  int dq_inline;  // ln_tanh_bwd_multi, rank-1 mode: dq[row] = 2 (q - y) inv_norm computed from the LossArgs of the launch
};
int ln_tanh_bwd(const LnBwdArgs& a, int D, hipStream_t stream);
// up to kMaxMulti LayerNorm backward passes of possibly different widths in one launch (+ the critic-loss rider when loss.on)
int ln_tanh_bwd_multi(const LnBwdArgs* a, int n, const LossArgs& loss, hipStream_t stream);

int colsum(const float* X, const float* Y, int groups, int rows_per_group, int D, float* out,
           long out_gstride, bool accumulate, hipStream_t stream);
struct Colsum3Args {
  const float *dg, *xhat, *dpre; int groups, rows_per_group, D;
  float *o_gamma, *o_beta, *o_bias; long gstride;
  int mode;   // 0: the three sums of a Dense->LN->tanh layer; 1: o_beta[g][j] = sum_r dg[r][j] only (bias gradient of a plain
              // Dense); 2: o_beta[0] = sum_r dg[r] (a vector's sum: rows_per_group elements, D = groups = 1)
};
constexpr int kMaxColsum = 8;
int colsum3_multi(const Colsum3Args* layers, int n, hipStream_t stream);
int colsum3(const float* dg, const float* xhat, const float* dpre, int groups, int rows_per_group, int D,
            float* o_gamma, float* o_beta, float* o_bias, long gstride, hipStream_t stream);
struct SleFwdArgs {
  const float* x; const float* K; const uint8_t* mask; float* f;
  // sle_proprio_fwd only: mask == nullptr && gen -> Dropout keep-mask hashed from (seed, camera, GLOBAL row, channel)
  // gen == 2: jax.random.bernoulli(tf_key[camera], keep, (tf_rows, channels * 8))[tf_row0 + row][...] instead of the hash (jaxrng.h)
  int gen; uint64_t seed; long row_offset, rows_global;
  uint32_t tf_key[4][2]; long tf_rows, tf_row0;
};
// SpatialLearnedEmbeddings channel-blocked (a workgroup = 256 channels x 8 samples: the kernel K is read once per workgroup,
// not once per sample) + inline Dropout mask + the proprio branch as extra workgroups of the same launch (pv == nullptr: none)
struct ProprioArgs;
int sle_proprio_fwd_multi(const SleFwdArgs* v, const ProprioArgs* pv, int n, float keep, int N, int HW, int Cc, int groups, long x_gs,
                          long k_gs, long mask_gs, long f_gs, int state_dim, hipStream_t stream);
int sle_fwd_multi(const SleFwdArgs* v, int n, float keep_scale, int N, int HW, int Cc, int groups, long x_gs, long k_gs,
                  long mask_gs, long f_gs, hipStream_t stream);
int sle_bwd(const float* x, const float* df, float* partial, int N, int HW, int Cc, int nsplit, int groups,
            long x_gs, long df_gs, long part_gs, hipStream_t stream);
// the same with the sum over the batch splits done by the last-arriving workgroup of every (camera, pixel, channel block):
// out[g][hw][c][j] (camera stride out_gs) -- no reduce_slabs launch.  ctr: groups * HW * cdiv(Cc, 256) zeroed counters.
int sle_bwd_fused(const float* x, const float* df, float* partial, int N, int HW, int Cc, int nsplit, int groups,
                  long x_gs, long df_gs, long part_gs, float* out, long out_gs, int* ctr, hipStream_t stream);
// dbias: gradient of the head bias -- one scalar (shared head) or, with per_member_bias, one per ensemble member.
// sel: the target ensemble members whose minimum backs up (sac.py:150-161: critic_subsample_size random members, n = 0: all).
// logp_next / alpha != nullptr: backup_entropy (sac.py:174-176): y -= alpha[0] * logp_next[b]
int critic_loss(const float* qt, const float* q, const float* reward, const float* mask, RedqSel sel, int E,
                int B, float discount, float inv_norm, float* y_out, float* dq, float* scalars, float* dbias,
                hipStream_t stream, bool per_member_bias = false, const float* logp_next = nullptr, const float* alpha = nullptr);
int policy_dist_fwd_multi(const PolicyDistArgs* v, int n, int B, int A, float std_min, float std_max, hipStream_t stream);
// proprio branch: y = tanh(LN(state W + b)) with W [S][64] (encoding.py:55-70), one wave per row
struct ProprioArgs {
  const float* state; const float *W, *b, *gamma, *beta;
  float* y; long ld_y; float* xhat; float* rstd;
  const float* copy_src; long ld_copy_src; float* copy_dst; long ld_copy_dst; int copy_cols;  // optional rider
};
int proprio_fwd_multi(const ProprioArgs* v, int n, int S, int rows, hipStream_t stream);
int policy_dist_bwd(const float* da, long ld_da, const float* act, long ld_act, const float* pre,
                    const float* stdv, const float* eps, const float* alpha, float coef, int B, int A,
                    float std_min, float std_max, float* dpre, const float* q, int E, float* qmean_out,
                    hipStream_t stream);  // rider: qmean_out[0] = sum_b mean_e q[e][b]
struct CopyJob { const float* src; long ld_src; float* dst; long ld_dst; int cols; };
int copy_cols_multi(const CopyJob* jobs, int n, int rows, hipStream_t stream);
int fill(float* p, float v, long n, hipStream_t stream);

struct AdamArgs {
  float *theta, *theta_target;
  long P, Pc, Pa0, Pa1;  // critic-tx support [0,Pc), actor-tx support [Pa0,Pa1), temperature = P-1
  const float *g_critic, *g_actor;
  float *m_c, *v_c, *m_a, *v_a, *m_t, *v_t;
  const float* sum_logp_next;  // device scalar: local/all-reduced sum of log pi(next)
  float* temp_grad_out;        // device scalar (debug/export)
  int critic_on, actor_on, temp_on;
  float lr_c, lr_a, lr_t, bc1, bc2, tau, target_entropy, inv_batch;
  // target EMA (common.py:124-134) runs when ema_on; optimizers with *_on == 0 step with a zero gradient
  int ema_on;
  // optional make_optimizer branches (optimizers.py:32-46); all zero = plain adam
  float wd_c, wd_a, wd_t;           // adamw weight decay of each optimizer (applies to EVERY leaf of the tree)
  float clip_c, clip_a, clip_t;     // clip_by_global_norm thresholds (<= 0: none)
  const float* norm2;               // device [2]: squared global norm of g_critic / g_actor (when a clip is active)
  // frozen (trunk) leaves appended to the index space: target EMA, and weight decay if any optimizer has one
  float* frozen; float* frozen_target; long n_frozen;
  long n_frozen_live; int vec_ok;   // set by adam_ema(): frozen leaves this launch covers; the 4-wide fast path may be used
  // info rider: 0 = none, 1 = critic step, 2 = actor/temperature step.  Slot order of `scalars` / `info_acc` as in
  // agent.hip (S_* / I_* enums); info_acc has 8 floats.
  int info_mode, info_reset;
  const float* scalars; const float* alpha; float* info_acc;
  float info_w, inv_eb;
};
int adam_ema(const AdamArgs& a, hipStream_t stream);
// `steps` target-EMA steps of frozen leaves in one pass (exactly the values `steps` adam_ema launches would have left)
int frozen_ema(const float* frozen, float* frozen_target, long n, float tau, long steps, hipStream_t stream);
// out[0] = sum g_critic^2 over [0, nc), out[1] = sum g_actor^2 over [0, na) (deterministic single-block reduction)
int grad_norm2(const float* g_critic, long nc, const float* g_actor, long na, float* out, hipStream_t stream);
// kind 0: N(0,1) f32, 1: keep-mask u8.  The tensor is [planes][rows_local][row_elems]; the value of an element is a
// hash of its position in the GLOBAL tensor [planes][rows_global][row_elems] (rows row_offset.. of it), so that a
// batch-sharded job draws the same noise for a sample whichever rank owns it (rows_global == 0: local == global)
struct NoiseJob { void* out; long n; uint64_t seed; int kind; float keep; long rows_local, rows_global, row_offset, row_elems; };
int gen_noise_multi(const NoiseJob* v, int n, hipStream_t stream);

}  // namespace serl
