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
 * f32[batch]; done_out u8[batch].  host_idx: int64[batch] (slot indices). */
int serl_rb_gather_packed(serl_rb* rb, const int64_t* host_idx, int batch,
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
 * host_crop_obs / host_crop_next: int32[batch][2] = (dy,dx) in [0,8]; NULL = no shift (4,4). */
int serl_rb_gather_crop(serl_rb* const* rbs, int n_rb, const int64_t* const* host_idx,
                        const int* counts, const int32_t* host_crop_obs,
                        const int32_t* host_crop_next, const serl_batch* out, void* stream);

/* _unpack + random shift on an already-gathered packed batch (train_utils.py:44-66,
 * data_augmentations.py:7-36): dev_packed[c] u8[batch][2][H][W][C] -> out->frames. */
int serl_crop_packed(int device, const uint8_t* const* dev_packed, int n_cam, int batch, int H,
                     int W, int C, const int32_t* host_crop_obs, const int32_t* host_crop_next,
                     uint8_t* dev_frames_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SERL_MI355_H */
