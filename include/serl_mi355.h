/* serl_mi355.h -- C ABI of libserl_mi355.so, the MI355X (gfx950) learner hot path for SERL.
 *
 * The reference (rail-berkeley/serl) is pure Python/JAX and has no FFI seam; the seam is the
 * Python object API used by the learner loop (examples/async_drq_sim/async_drq_sim.py:183-311).
 * Each entry point below replaces the reference method(s) cited next to it; INTEGRATION.md shows
 * the ctypes stub a maintainer would add on the reference side.  All paths are relative to
 * serl_launcher/serl_launcher/ in the reference tree.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; serl_last_error() gives the message
 *     (thread-local).  No exceptions, no Python or torch types cross this boundary.
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g. torch tensors'
 *     data_ptr()); "host" pointers are ordinary host memory.  `stream` is a hipStream_t passed
 *     as void* (0 = null stream).  Kernels are enqueued asynchronously on it.
 *   - handles own their device memory (hipMalloc) and are freed by the matching *_destroy.
 */
#ifndef SERL_MI355_H
#define SERL_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SERL_OK 0
#define SERL_ERR_INVALID (-1)
#define SERL_ERR_HIP (-2)
#define SERL_ERR_STATE (-3)
#define SERL_ERR_UNSUPPORTED (-4)

#define SERL_MAX_CAMS 4
#define SERL_MAX_BUFFERS 2

const char* serl_last_error(void);
int serl_version(void);
/* number of visible HIP devices (0 if none); does not fail without a GPU */
int serl_device_count(void);

/* HIP-event timing of the hot kernels (off by default).  serl_profile_enable(k): k = 0 off, k >= 1 times every
 * k-th launch of each instrumented tag (an event pair costs a few microseconds on the stream, which matters
 * once the step itself is below a millisecond).  serl_profile_read returns, per tag, the summed duration and
 * the number of TIMED launches since the last reset:
 * names is char[max_entries][64]. Used by bench.py for the live roofline numbers. */
int serl_profile_enable(int on);
int serl_profile_reset(void);
int serl_profile_read(int max_entries, char* names, double* total_ms, int64_t* counts, int* n_out);

/* ------------------------------------------------------------------------------------------
 * Replay buffer  (data/memory_efficient_replay_buffer.py, data/replay_buffer.py, data/dataset.py)
 * Frames live in HBM: one u8[H*W*C] frame per slot per camera (the *next* frame of the
 * transition; observation frame of slot i is slot i-1), plus one f32 record per slot.
 * ------------------------------------------------------------------------------------------ */
typedef struct serl_rb serl_rb;

/* MemoryEfficientReplayBuffer.__init__ (memory_efficient_replay_buffer.py:13-51) +
 * ReplayBuffer.__init__ (replay_buffer.py:41-66). num_stack = T (frames per observation). */
int serl_rb_create(int device, int64_t capacity, int n_cam, int H, int W, int C, int num_stack,
                   int state_dim, int act_dim, serl_rb** out);
int serl_rb_destroy(serl_rb* rb);

/* Dataset.seed (dataset.py:72-74): the caller builds numpy's PCG64(SeedSequence(seed)) and hands
 * over its raw state (bit_generator.state), so SeedSequence hashing stays in numpy. */
int serl_rb_seed(serl_rb* rb, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi,
                 uint64_t inc_lo, int has_uint32, uint32_t uinteger);
/* read back the generator state (tests: must equal numpy's after the same draws) */
int serl_rb_rng_state(serl_rb* rb, uint64_t out_state_inc[4], int* has_uint32, uint32_t* uinteger);

/* MemoryEfficientReplayBuffer.insert (memory_efficient_replay_buffer.py:53-89), thread-safe like
 * MemoryEfficientReplayBufferDataStore.insert (data_store.py:104-106).
 * obs_frames[c] / next_frames[c]: host u8[T][H][W][C] for camera c; state/next_state: host
 * f32[T*S]; action: host f32[A]. */
int serl_rb_insert(serl_rb* rb, const uint8_t* const* obs_frames,
                   const uint8_t* const* next_frames, const float* state,
                   const float* next_state, const float* action, float reward, float mask,
                   int done);

int64_t serl_rb_len(serl_rb* rb);          /* ReplayBuffer.__len__ (replay_buffer.py:68-69) */
int64_t serl_rb_insert_index(serl_rb* rb); /* latest_data_id (data_store.py:138-140) */
int serl_rb_valid_mask(serl_rb* rb, uint8_t* host_out /* [capacity] */);

/* index draw + rejection loop (memory_efficient_replay_buffer.py:111-122): bit-exact with
 * numpy Generator(PCG64).integers.  host_idx_out: int64[batch]. */
int serl_rb_sample_indices(serl_rb* rb, int batch, int64_t* host_idx_out);

/* gather of sample(pack_obs_and_next_obs=True) (memory_efficient_replay_buffer.py:126-164,
 * dataset.py:40-51).  dev outputs:  frames_out[c] u8[batch][T+1][H][W][C];
 * state_out/next_state_out f32[batch][T*S]; action_out f32[batch][A]; reward_out, mask_out
 * f32[batch]; done_out u8[batch].  host_idx: int64[batch] slot indices, IN/OUT: an index whose slot an insert has invalidated
 * since it was drawn is re-drawn (from the buffer's generator) and written back, so the array always describes the batch
 * that was gathered.  Gathers may be issued from several streams; an insert that overwrites a slot waits for all of them. */
int serl_rb_gather_packed(serl_rb* rb, int64_t* host_idx, int batch,
                          uint8_t* const* dev_frames_out, float* dev_state_out,
                          float* dev_next_state_out, float* dev_action_out,
                          float* dev_reward_out, float* dev_mask_out, uint8_t* dev_done_out,
                          void* stream);

/* Device batch consumed by the agent: the result of sample -> [concat_batches] -> _unpack ->
 * random-shift crop.  All pointers are device addresses owned by the caller.
 *   frames : u8 [2 (0=obs,1=next)][n_cam][batch][H][W][C]   (T==1)
 *   state  : f32[2][batch][S]
 */
typedef struct serl_batch {
  int batch, n_cam, H, W, C, state_dim, act_dim;
  uint8_t* frames;
  float* state;
  float* action;
  float* reward;
  float* mask;
  uint8_t* done;
} serl_batch;

/* Fused sample-gather + concat_batches + _unpack + DrQ random shift (K2+K3+K4 of SURVEY.md):
 * memory_efficient_replay_buffer.py:126-164 + utils/train_utils.py:16-31,44-66 +
 * vision/data_augmentations.py:7-36 + agents/continuous/drq.py:244-253 (same offsets for every
 * camera).  Samples [0,counts[0]) come from rbs[0], the next counts[1] from rbs[1] (RLPD 50/50).
 * host_crop_obs / host_crop_next: int32[batch][2] = (dy,dx) in [0,8]; NULL = no shift (4,4).
 * host_idx[b]: IN/OUT like serl_rb_gather_packed's (stale indices are re-drawn in place). */
int serl_rb_gather_crop(serl_rb* const* rbs, int n_rb, int64_t* const* host_idx,
                        const int* counts, const int32_t* host_crop_obs,
                        const int32_t* host_crop_next, const serl_batch* out, void* stream);

/* _unpack + random shift on an already-gathered packed batch (train_utils.py:44-66,
 * data_augmentations.py:7-36): dev_packed[c] u8[batch][2][H][W][C] -> out->frames. */
int serl_crop_packed(int device, const uint8_t* const* dev_packed, int n_cam, int batch, int H,
                     int W, int C, const int32_t* host_crop_obs, const int32_t* host_crop_next,
                     uint8_t* dev_frames_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * DrQ / SAC agent  (agents/continuous/drq.py, agents/continuous/sac.py, common/common.py)
 * One handle = JaxRLTrainState (params, target_params, 3 Adam states, step) + all activations.
 * ------------------------------------------------------------------------------------------ */
typedef struct serl_agent serl_agent;

/* hyper-parameters of make_drq_agent / DrQAgent.create_drq (utils/launcher.py:79-116,
 * agents/continuous/drq.py:24-242) with encoder_type == "resnet-pretrained". */
typedef struct serl_agent_cfg {
  int device;
  int n_cam, H, W;         /* image_keys order is the caller's; C == 3.  n_cam == 0: state-only SAC
                            * (SACAgent.create_states, sac.py:486-542): no encoder, observations are the flat
                            * state vectors, one Dense(1) Q head per ensemble member; H, W ignored */
  int state_dim, act_dim;
  int batch;               /* max samples per update call on THIS rank */
  int ensemble;            /* critic_ensemble_size (10) */
  int hidden;              /* MLP width (256) */
  int bottleneck;          /* encoder bottleneck_dim (256) */
  int sle_features;        /* num_spatial_blocks (8) */
  int proprio_dim;         /* proprio_latent_dim (64) */
  int warmup_steps;        /* optimizers.py:23-30: actor and critic optimizers (0 for DrQ, 2000 for make_sac_agent) */
  int temp_warmup_steps;   /* temperature optimizer's own warm-up; < 0: same as warmup_steps (sac.py:333-343: none) */
  float discount, tau, lr;
  float dropout;           /* 0.1, resnet_v1.py:351 */
  float std_min, std_max;  /* 1e-5, 5 */
  float target_entropy;    /* -act_dim/2, drq.py:88-89 */
  uint64_t seed;           /* device noise stream (production mode) */
  /* Per-optimizer options of make_optimizer (common/optimizers.py:6-56), indexed by SERL_TX_*.  All zero (the
   * default, what every example uses) = adam + the warm-up -> constant schedule above.
   *   tx_lr            > 0: learning rate of this optimizer (else `lr`); with tx_lr_set[t] != 0 the value is taken as
   *                    given, 0 included (optax accepts learning_rate=0.0, e.g. to freeze the temperature)
   *   tx_warmup       >= 0: warm-up steps of this optimizer, given as steps + 1 (0 = use warmup_steps / temp_warmup_steps)
   *   tx_cosine_steps  > 0: warmup_cosine_decay_schedule(0, lr, warmup, decay_steps = this, end 0) (optimizers.py:14-21)
   *   tx_weight_decay_on != 0: optax.adamw with tx_weight_decay (optimizers.py:39-42); it decays the WHOLE tree, as the
   *                    reference's tx.update(grads, state, params) does.  State-only agents (n_cam == 0) only: on a
   *                    pixel agent it would decay the frozen pretrained trunk (SERL_ERR_UNSUPPORTED)
   *   tx_clip_norm     > 0: optax.clip_by_global_norm on this optimizer's gradient tree (optimizers.py:36-37) */
  float tx_lr[3];
  int tx_warmup[3];
  int tx_cosine_steps[3];
  int tx_weight_decay_on[3];
  float tx_weight_decay[3];
  float tx_clip_norm[3];
  /* encoder_type of DrQAgent.create_drq (drq.py:137-186): SERL_ENCODER_RESNET_PRETRAINED (0, the frozen ResNet-10 of
   * every example) or SERL_ENCODER_SMALL: the trainable SmallEncoder (vision/small_encoders.py:9-55, features
   * (32,64,128,256), 3x3 stride-2 VALID convs + ReLU, average pool, Dense(256)+LayerNorm+tanh), gradients of the
   * critic loss flow into its conv kernels */
  int encoder_type;
  int critic_subsample_size; /* sac.py:150-161: 0 = 2 (utils/launcher.py), -1 = None (minimum over all members), else 1..16 */
  int backup_entropy;        /* sac.py:174-176: target_q -= alpha * log pi(a'|s') */
  int tx_lr_set[3];          /* != 0: tx_lr[t] was given explicitly and is honoured even when it is 0 */
} serl_agent_cfg;
#define SERL_ENCODER_RESNET_PRETRAINED 0
#define SERL_ENCODER_SMALL 1
#define SERL_TX_ACTOR 0
#define SERL_TX_CRITIC 1
#define SERL_TX_TEMPERATURE 2

int serl_agent_create(const serl_agent_cfg* cfg, serl_agent** out);
int serl_agent_destroy(serl_agent* a);

/* Parameter tree access (flat leaf names, see DESIGN.md "parameter arena"; the Python shim maps
 * them to the flax tree of agent.state.params for publish_network / checkpoints).
 * section: "params" | "target_params" | "opt/<tx>/mu" | "opt/<tx>/nu", tx in actor|critic|temperature */
int serl_agent_num_leaves(serl_agent* a);
int serl_agent_leaf_info(serl_agent* a, int i, char* name_out, int name_cap, int64_t* count);
int serl_agent_set(serl_agent* a, const char* section, const char* leaf, const float* host, int64_t count);
int serl_agent_get(serl_agent* a, const char* section, const char* leaf, float* host_out, int64_t count);
int serl_agent_set_step(serl_agent* a, int64_t step); /* JaxRLTrainState.step and the Adam counts */
/* Arithmetic of the frozen trunk's 3x3/1x1 convs: 0 = exact fp32 MFMA, 1 (default) = split-fp16
 * ("f16x3": x = hi + 2^-11 lo', three fp16 MFMA products per fp32 product, fp32 accumulate; error per
 * product <= ~3*2^-22, i.e. fp32-roundoff class -- DESIGN.md section 4; parity tests run both modes). */
int serl_agent_set_trunk_mode(serl_agent* a, int mode);
/* Scheduling hint (results equal up to fp32 summation order: the K-split depth sets how many partial sums a GEMM adds;
 * per agent, read by that agent's launches only): the most workgroups a K-split GEMM launch of the update (sac.py:243-299) may use
 * (0 = default 512).  A caller that overlaps the update of batch i with the frozen-trunk pass of batch i+1 at a large per-rank
 * batch sets 256: fewer update workgroups queue for CU slots between the trunk's conv workgroups. */
int serl_agent_set_chain_budget(serl_agent* a, int workgroups);
int64_t serl_agent_get_step(serl_agent* a);

/* Explicit randomness (parity mode).  All pointers are DEVICE addresses; NULL members (or a NULL
 * struct) are generated on the device from cfg.seed.  Shapes use the batch of the call:
 *   eps_*  f32[batch][act_dim];  mask_* u8[n_cam][batch][512*sle_features] keep-masks (1 = keep)
 * redq_idx is a HOST pointer: int32[utd_ratio][critic_subsample_size] (sac.py:150-157; unused when the size is None). */
typedef struct serl_noise {
  const float* eps_next;          /* critic loss: policy sample at next_obs (sac.py:118-132) */
  const uint8_t* mask_next;       /* Dropout(0.1) masks of that policy forward */
  const int32_t* redq_idx;        /* host */
  const float* eps_pi;            /* policy loss sample at obs (sac.py:197-201) */
  const uint8_t* mask_obs_pi;
  const float* eps_temp;          /* temperature loss sample at next_obs (sac.py:224-227) */
  const uint8_t* mask_next_temp;
  /* Round 5 -- jax.random KEYS instead of tensors (HOST pointers to uint32 words; NULL = none).  Where the tensor above is NULL
   * and its key is given, the draws are jax.random's for that key (serl_jax_* below: bits exact, normals through XLA's float32
   * erf_inv), produced INSIDE the kernels that consume them -- no noise tensor exists, no extra launch.  Array shapes are the
   * reference's: normal(key, (global minibatch rows, act_dim)), bernoulli(key_cam, 1 - dropout, (global minibatch rows, 4096));
   * a rank of a data-parallel job draws its rows of them (serl_agent_set_shard).  key_*_next are indexed by the minibatch of a
   * high-UTD update (redq_row): [utd][2] and [utd][n_cam][2]; the others are [2] and [n_cam][2]. */
  const uint32_t* key_eps_next;
  const uint32_t* key_mask_next;
  const uint32_t* key_eps_pi;
  const uint32_t* key_mask_obs_pi;
  const uint32_t* key_eps_temp;
  const uint32_t* key_mask_next_temp;
} serl_noise;

/* sac.py:185-189,215-219,234,291-297 */
typedef struct serl_info {
  float critic_loss, predicted_qs, target_qs;
  float actor_loss, temperature, entropy, temperature_loss;
  float actor_lr, critic_lr, temperature_lr;
} serl_info;

/* DrQAgent.update_critics (drq.py:296-328) on an augmented device batch: one critic grad-step
 * (REDQ target, 10-member ensemble MSE), 3x Adam (actor/temperature with zero gradients), target EMA. */
int serl_agent_update_critics(serl_agent* a, const serl_batch* batch, const serl_noise* noise, void* stream);
/* DrQAgent.update_high_utd (drq.py:255-294 -> sac.py:544-596): utd_ratio critic updates on
 * consecutive minibatches, then one actor + temperature update on the full batch. */
int serl_agent_update_high_utd(serl_agent* a, const serl_batch* batch, int utd_ratio,
                               const serl_noise* noise, void* stream);
/* info of the last update call (synchronises `stream`) */
int serl_agent_read_info(serl_agent* a, serl_info* host_out, void* stream);

/* Data-parallel phases (common.py:213-214 lax.pmean made real).  Rank r holds samples
 * [r*B/P,(r+1)*B/P) of the global batch; the caller all-reduces (SUM) the gradient view between
 * *_grads and apply.  global_count = samples of this (mini)batch over all ranks (loss normaliser). */
int serl_agent_encode(serl_agent* a, const serl_batch* batch, void* stream); /* frozen trunk, both passes */
/* Software pipelining: the frozen trunk does not depend on the trainable parameters, so the trunk
 * pass of batch i+1 (another slot, on a second stream) may overlap the update of batch i.  There are
 * THREE slots (0..2): with two, the pass of batch i+2 has to wait ON THE DEVICE for the update of batch i
 * (a cross-stream dependency, 60-100 us on this stack); with three it reuses the slot of batch i-1, whose
 * update a host that runs one step ahead can confirm with an event query.  The caller orders the
 * streams with events; serl_agent_select_slot picks the batch the following *_grads calls consume.
 * serl_agent_encode == encode_slot(0) + select_slot(0). */
int serl_agent_encode_slot(serl_agent* a, const serl_batch* batch, int slot, void* stream);
/* The same pass issued in consecutive pieces: stages [stage_begin, stage_end] with -1 = conv_init + max-pool and 0..3 = the
 * residual stages (split-fp16 trunk, full-size batch).  The caller may record an event between two pieces, e.g. to start
 * the update of the previous batch only once this pass has left its first stages. */
int serl_agent_encode_slot_range(serl_agent* a, const serl_batch* batch, int slot, int stage_begin, int stage_end, void* stream);
int serl_agent_select_slot(serl_agent* a, int slot);
/* Features computed ELSEWHERE.  The trunk is frozen and its output is cut off by a stop_gradient (vision/resnet_v1.py:286), so
 * the features of a batch depend on its pixels only -- another GPU can run that pass ("trunk farm": serl_amd/parallel.py
 * TrunkFarmLearner).  serl_agent_slot_features gives the slot's feature buffer f32[2][n_cam][batch][h*w][512] (what encode_slot
 * writes; a producer sends from it, the updating rank receives into it); serl_agent_bind_slot attaches a batch (states, actions,
 * rewards, masks -- and its frames, which the update does not read) to a slot WITHOUT running the trunk. */
int serl_agent_slot_features(serl_agent* a, int slot, float** dev_out, int64_t* count_out);
int serl_agent_bind_slot(serl_agent* a, const serl_batch* batch, int slot);
int serl_agent_critic_grads(serl_agent* a, int offset, int count, int global_count,
                            const serl_noise* noise, int redq_row, void* stream);
/* The same critic phase with its gradients published in two BUCKETS so the caller can all-reduce the first while the
 * second is still being computed (the pmean of common.py:213-214 overlapped with the backward pass, as DDP-style bucketing
 * does): `event_bucket0` (a hipEvent_t, may be NULL) is recorded on `stream` as soon as bucket 0 is final -- before the
 * encoder-head backward (Dense 4096->256 weight gradient, SpatialLearnedEmbeddings / SmallEncoder gradients) is issued.
 *   bucket 0 = [critic ensemble | Q head | proprio branch | loss scalars]   (contiguous; ~8.7 MB at the headline shape)
 *   bucket 1 = [per-camera encoder heads]                                    (contiguous; ~8.9 MB; empty for state-only SAC)
 * Both are final when the call's work on `stream` has completed.  serl_agent_grad_bucket returns their device ranges. */
int serl_agent_critic_grads_bucketed(serl_agent* a, int offset, int count, int global_count, const serl_noise* noise,
                                     int redq_row, void* stream, void* event_bucket0);
int serl_agent_grad_bucket(serl_agent* a, int bucket, float** dev_ptr, int64_t* count);
int serl_agent_actor_grads(serl_agent* a, int global_count, const serl_noise* noise, void* stream);
/* `which` = networks_to_update of SACAgent.update (sac.py:243-299) as a bit set; optimizers whose bit is clear still
 * step with a zero gradient (sac.py:276-277).  The target EMA runs iff SERL_NET_CRITIC is set (sac.py:284-285). */
#define SERL_NET_CRITIC 1
#define SERL_NET_ACTOR 2
#define SERL_NET_TEMPERATURE 4
#define SERL_APPLY_CRITIC SERL_NET_CRITIC
#define SERL_APPLY_ACTOR_TEMP (SERL_NET_ACTOR | SERL_NET_TEMPERATURE)
int serl_agent_apply(serl_agent* a, int which, float info_weight, void* stream);
/* SACAgent.update(batch, networks_to_update) (sac.py:243-299) on an augmented device batch: every selected loss is
 * evaluated at the SAME (pre-update) parameters, then ONE optimizer step of all three Adam transforms (+ target EMA
 * if the critic is selected).  nets = any non-empty combination of SERL_NET_*. */
int serl_agent_update(serl_agent* a, const serl_batch* batch, int nets, const serl_noise* noise, void* stream);
int serl_agent_begin_update(serl_agent* a, void* stream); /* clears the info accumulator */
/* Batch-sharded data parallelism: the batches this agent is given are rows [global_offset, global_offset + local
 * batch) of a global batch of `global_batch` rows.  Device-generated noise (noise == NULL) is then indexed by the
 * GLOBAL row, so a sample gets the same eps / dropout mask whichever rank owns it and results do not depend on the
 * world size (the dormant pmean of common.py:213-214 made real).  global_batch = 0: not sharded (default). */
int serl_agent_set_shard(serl_agent* a, int64_t global_offset, int64_t global_batch);
/* which = SERL_APPLY_CRITIC: [critic grads | scalars]; SERL_APPLY_ACTOR_TEMP (or any actor/temperature bit):
 * [scalars | actor grads]; critic AND actor/temperature bits: the whole [critic grads | scalars | actor grads] range */
int serl_agent_grad_view(serl_agent* a, int which, float** dev_ptr, int64_t* count);

/* SACAgent.sample_actions (sac.py:301-320): policy forward with train=False on `n` observations
 * (frames u8[n_cam][n][H][W][3], state f32[n][S], both device); eps f32[n][A] device or NULL for
 * argmax (= distribution mode).  out_actions f32[n][A] device. */
int serl_agent_sample_actions(serl_agent* a, const uint8_t* dev_frames, const float* dev_state, int n,
                              const float* dev_eps, float* dev_out_actions, void* stream);

/* Debug / test taps (device->host copies of internal activations). */
int serl_agent_trunk_forward(serl_agent* a, const uint8_t* dev_frames, int n, float* dev_feats_out, void* stream);
int serl_agent_debug_get(serl_agent* a, const char* what, float* host_out, int64_t count);
/* inject gradients / scalars ("g_critic", "g_actor", "scalars") before serl_agent_apply: optimizer tests */
int serl_agent_debug_set(serl_agent* a, const char* what, const float* host, int64_t count);
/* Which kernels the LAST split-fp16 trunk pass selected, as text ("images=128 pool=1 raw_b0=0 b0_conv0=S/1/0/f1 ...":
 * layer=kernel/tile-config/statistics-mode/f<fused epilogue>; kernel S = row-slab, D = LDS-DMA, R = register-staged implicit
 * GEMM).  The per-rank shapes of a data-parallel job (resnet_v1.py:260-269 at N = B/8 images) pick other kernels than the
 * full batch; the parity tests assert which path they exercised. */
int serl_agent_trunk_plan(serl_agent* a, char* out, int cap);
/* Number of update-chain kernels (everything but the frozen trunk, the SmallEncoder convs and the replay kernels) launched
 * since the library was loaded: the tests pin the launch count of an update_critics + update_high_utd pair (sac.py:243-299). */
int64_t serl_debug_chain_launches(void);

/* ---------------------------------------------------------------------------------------------
 * Reward classifier, inference only (next-row N4; serl_launcher/networks/reward_classifier.py:16-113).
 * BinaryClassifier = EncodingWrapper(use_proprio=False) over the frozen ResNet-10 trunk (per camera:
 * SpatialLearnedEmbeddings(8) -> Dense(256) -> LayerNorm -> tanh, concatenated) -> Dense(256) -> [Dropout: identity at
 * train=False] -> LayerNorm -> ReLU -> Dense(1).  serl_classifier_logits is the function load_classifier_func
 * (reward_classifier.py:93-113) returns: observations in, logits out.  Leaves (flat fp32, HWIO kernels / [in][out]
 * dense kernels): the trunk leaves of the agent ("trunk/..."), "enc/<k>/{sle, dense/kernel, dense/bias, ln/scale,
 * ln/bias}", "head/dense0/{kernel,bias}", "head/ln/{scale,bias}", "head/dense1/{kernel,bias}".
 * --------------------------------------------------------------------------------------------- */
typedef struct serl_classifier serl_classifier;
typedef struct {
  int device, n_cam, H, W, max_batch;
} serl_classifier_cfg;
int serl_classifier_create(const serl_classifier_cfg* cfg, serl_classifier** out);
int serl_classifier_destroy(serl_classifier* c);
int serl_classifier_num_leaves(serl_classifier* c);
int serl_classifier_leaf_info(serl_classifier* c, int i, char* name_out, int name_cap, int64_t* count);
int serl_classifier_set(serl_classifier* c, const char* leaf, const float* host, int64_t count);
int serl_classifier_get(serl_classifier* c, const char* leaf, float* host_out, int64_t count);
/* dev_frames u8[n_cam][n][H][W][3] (device), n <= max_batch; dev_logits f32[n] (device) */
int serl_classifier_logits(serl_classifier* c, const uint8_t* dev_frames, int n, float* dev_logits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * JAX's PRNG (threefry2x32, non-partitionable; jax/_src/prng.py, jax/_src/random.py of jax 0.4.x) -- csrc/jaxrng.hip.
 * The reference learner draws its crop offsets (vision/data_augmentations.py:7-36), REDQ indices (agents/continuous/sac.py:150-157),
 * policy noise (sac.py:118-132,197-201,224-227) and Dropout masks from jax.random and advances `state.rng` as
 * common/common.py:197-209, sac.py:287-289 and agents/continuous/drq.py:276-318 do.  These entry points reproduce that stream:
 * keys, integers and the 32-bit draws behind every sample are bit-exact; normals follow XLA's float32 erf_inv polynomial.
 * A key is uint32[2]; jax.random.PRNGKey(seed) = {0, seed & 0xffffffff} as the reference runs JAX (x64 disabled: the seed is an
 * int32, PRNGKey(-1) = {0, 0xffffffff}).  The host functions need no device.
 * --------------------------------------------------------------------------------------------- */
int serl_jax_prngkey(uint64_t seed, uint32_t key_out[2]);                              /* jax.random.PRNGKey */
int serl_jax_split(const uint32_t key[2], int num, uint32_t* keys_out /* [num][2] */); /* jax.random.split */
int serl_jax_fold_in(const uint32_t key[2], uint32_t data, uint32_t key_out[2]);       /* jax.random.fold_in */
int serl_jax_random_bits(const uint32_t key[2], int64_t n, uint32_t* out);             /* jax.random.bits(key, (n,)) */
int serl_jax_randint(const uint32_t key[2], int64_t n, int32_t minval, int32_t maxval, int32_t* out); /* jax.random.randint */
int serl_jax_normal_host(const uint32_t key[2], int64_t n, float* out);                /* jax.random.normal(key, (n,)) on the host */
/* batched_random_crop's offsets (data_augmentations.py:22-36): keys = split(key, frames); (y, x)_i = randint(keys[i], (2,), 0,
 * 2*padding + 1) -> yx_out int32[frames][2].  The SAME key serves every camera (drq.py:244-253). */
int serl_jax_crop_offsets(const uint32_t key[2], int frames, int padding, int32_t* yx_out);
/* Every key ONE learner call derives from state.rng, in the reference's order:
 *   drq_aug != 0 (DrQAgent.update_critics / update_high_utd, drq.py:276-277,307-308): rng, k_obs, k_next = split(rng, 3)
 *   then n_critic critic-only SACAgent.update calls and, if has_actor_temp, one actor + temperature update (sac.py:544-596); each:
 *     _, r_actor, r_critic, r_temp = split(rng, 4)                                          (common.py:197-200, sorted loss names)
 *     critic:  c, k_next_action = split(r_critic);  _, k_subsample = split(c)               (sac.py:137,151)
 *     actor:   _, k_policy, k_sample, _ = split(r_actor, 4);  _, k_temp = split(r_temp)     (sac.py:197,222)
 *     rng = split(rng)[0]                                                                   (sac.py:287-289)
 * k_next_action / k_policy / k_temp are BOTH the Dropout rng of that policy forward and (k_next_action, k_temp) its sample seed
 * (sac.py:122-128); k_sample seeds the policy-loss sample.  rng_out is state.rng after the call.
 * combined != 0 (SACAgent.update with all three networks, sac.py:243-299): ONE update whose critic, actor and temperature losses
 * take their keys from the same 4-way split (n_critic = 1, has_actor_temp = 1). */
#define SERL_JAX_MAX_UTD 32
typedef struct serl_jax_update_keys_t {
  uint32_t rng_out[2];
  uint32_t k_obs[2], k_next[2];
  int32_t n_critic;
  uint32_t k_next_action[SERL_JAX_MAX_UTD][2];
  uint32_t k_subsample[SERL_JAX_MAX_UTD][2];
  uint32_t k_policy[2], k_sample[2], k_temp[2];
} serl_jax_update_keys_t;
int serl_jax_update_keys(const uint32_t rng[2], int drq_aug, int n_critic, int has_actor_temp, int combined, serl_jax_update_keys_t* out);
/* Device draws: job i writes elements [first, first + count) of the flat array a JAX call of `n_total` elements returns
 * (a rank of a data-parallel job, or a minibatch window, fills its rows only):
 *   SERL_JAX_NORMAL       f32  jax.random.normal(key, shape)             SERL_JAX_BERNOULLI_U8  u8  jax.random.bernoulli(key, p, shape)
 *   SERL_JAX_BITS         u32  jax.random.bits(key, shape)
 * One launch for up to 16 jobs on `stream`. */
enum { SERL_JAX_NORMAL = 0, SERL_JAX_BERNOULLI_U8 = 1, SERL_JAX_BITS = 2 };
typedef struct serl_jax_job {
  uint32_t key[2];
  int32_t kind;
  float p;              /* bernoulli probability of a 1 */
  int64_t n_total, first, count;
  void* out;            /* device */
} serl_jax_job;
int serl_jax_fill(int device, const serl_jax_job* jobs, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SERL_MI355_H */
