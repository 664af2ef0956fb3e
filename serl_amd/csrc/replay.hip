// HBM-resident memory-efficient replay buffer for SERL on MI355X (gfx950).
//
// Host side: slot bookkeeping identical to the reference's MemoryEfficientReplayBuffer
// (serl_launcher/data/memory_efficient_replay_buffer.py:53-89) and a bit-exact PCG64/Lemire index
// sampler (numpy Generator.integers; memory_efficient_replay_buffer.py:111-122).
// Device side: one fused kernel = sample gather + concat_batches + _unpack + DrQ random shift
// (K2+K3+K4).  It is HBM-bound u8 traffic: every source frame row is pulled once with 16-byte
// coalesced loads into LDS, shifted/clamped out of LDS, and written once with 16-byte stores.
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "prof.h"

namespace serl {

thread_local char g_err[512] = {0};
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// PCG64 XSL-RR 128/64 with numpy's buffered next_uint32 and Lemire bounded draws.
// ---------------------------------------------------------------------------------------------
struct Pcg64 {
  unsigned __int128 state = 0, inc = 0;
  int has_uint32 = 0;
  uint32_t uinteger = 0;
  bool seeded = false;

  uint64_t next64() {
    const unsigned __int128 mult =
        ((unsigned __int128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;
    state = state * mult + inc;
    uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    uint64_t x = hi ^ lo;
    unsigned r = (unsigned)(state >> 122);
    return (x >> r) | (x << ((64 - r) & 63));
  }
  uint32_t next32() {
    if (has_uint32) {
      has_uint32 = 0;
      return uinteger;
    }
    uint64_t v = next64();
    has_uint32 = 1;
    uinteger = (uint32_t)(v >> 32);
    return (uint32_t)v;
  }
  // Generator.integers(n), 0 < n < 2^32 - 1 (numpy buffered_bounded_lemire_uint32)
  uint32_t bounded(uint32_t n) {
    if (n == 1) return 0;
    uint64_t m = (uint64_t)next32() * n;
    uint32_t l = (uint32_t)m;
    if (l < n) {
      uint32_t t = (0xFFFFFFFFu - (n - 1)) % n;
      while (l < t) {
        m = (uint64_t)next32() * n;
        l = (uint32_t)m;
      }
    }
    return (uint32_t)(m >> 32);
  }
};

constexpr int kRing = 8;           // staging slots for per-call index/crop parameters
constexpr int kInsRing = 32;       // pinned staging slots for inserted transitions (one slot write each)
constexpr int kRowsPerBlock = 32;  // output rows per workgroup in the gather/crop kernel

}  // namespace serl

struct serl_rb {
  int device = 0;
  int64_t cap = 0;
  int n_cam = 0, H = 0, W = 0, C = 0, T = 1, S = 0, A = 0;
  int rec_len = 0;  // floats per record: [state T*S | next_state T*S | action A | reward | mask | done]
  size_t frame_bytes = 0;
  uint8_t* frames[SERL_MAX_CAMS] = {nullptr};  // device, [cap][H*W*C] each
  float* rec = nullptr;                        // device, [cap][rec_len]
  // host bookkeeping (memory_efficient_replay_buffer.py)
  std::vector<uint8_t> valid;
  std::vector<float> rec_host;  // host mirror of the records (needed for the wrap re-insert)
  int64_t size = 0, insert_index = 0;
  bool first = true;
  serl::Pcg64 rng;
  std::mutex mu;
  // stream/event plumbing
  hipStream_t copy_stream = nullptr;
  // one "last gather" event per stream that gathers from this buffer (the learner's update stream, its prefetch side
  // stream, ...): an overwriting insert waits for ALL of them
  static constexpr int kGatherStreams = 4;
  hipEvent_t gather_ev[kGatherStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t gather_stream[kGatherStreams] = {nullptr, nullptr, nullptr, nullptr};
  bool gather_used[kGatherStreams] = {false, false, false, false};
  bool gather_pend[kGatherStreams] = {false, false, false, false};
  int gather_rr = 0;
  bool gather_pending = false;
  // inserts are staged through a pinned ring and copied asynchronously on copy_stream: the caller's thread (the
  // actor-facing server thread of data_store.py:104-106) holds the mutex for a host memcpy only, never for a stream
  // synchronisation, so it cannot stall the learner thread's sample_indices / gather
  uint8_t* ins_host = nullptr;
  size_t ins_slot_bytes = 0;
  hipEvent_t ins_done[serl::kInsRing] = {nullptr};
  bool ins_used[serl::kInsRing] = {false};
  int ins_next = 0;
  hipEvent_t last_insert = nullptr;
  bool insert_pending = false;
  // per-call parameter staging (pinned host + device), ring of kRing slots
  uint8_t* stage_host = nullptr;
  uint8_t* stage_dev = nullptr;
  size_t stage_slot_bytes = 0;
  hipEvent_t stage_done[serl::kRing] = {nullptr};
  bool stage_used[serl::kRing] = {false};
  int stage_next = 0;
};

namespace serl {

// ---------------------------------------------------------------------------------------------
// device kernels
// ---------------------------------------------------------------------------------------------
struct GatherArgs {
  const uint8_t* frames[SERL_MAX_BUFFERS][SERL_MAX_CAMS];
  const float* rec[SERL_MAX_BUFFERS];
  const int64_t* idx[SERL_MAX_BUFFERS];  // device, per buffer
  int64_t cap[SERL_MAX_BUFFERS];
  int count0;                            // samples [0,count0) come from buffer 0
  int batch, n_cam, H, W, C, S, A, rec_len;
  const int32_t* crop_obs;   // device [batch][2] or nullptr
  const int32_t* crop_next;  // device [batch][2] or nullptr
  uint8_t* out_frames;       // [2][n_cam][batch][H][W][C]
  float* out_state;          // [2][batch][S]
  float* out_action;         // [batch][A]
  float* out_reward;
  float* out_mask;
  uint8_t* out_done;
  int n_frame_blocks;
  // packed mode (serl_crop_packed): source is dev_packed[c] u8[batch][2][H][W][C]
  const uint8_t* packed[SERL_MAX_CAMS];
  int from_packed;
};

// Shift one output row out of an LDS-staged source row.  rowb = W*C bytes (multiple of 16).
// Interior 16-byte chunks: 5 aligned dword LDS reads + v_alignbyte; edge chunks (where the shift
// clamps to the border pixel) are assembled per byte.
template <int CT>
__device__ __forceinline__ uint4 shifted_chunk(const uint8_t* srow, int q, int sx, int W, int Crt) {
  const int C = CT > 0 ? CT : Crt;  // compile-time channel count (3) turns the /C, %C below into shifts/mults
  const int o0 = q * 16;
  const int pmin = o0 / C, pmax = (o0 + 15) / C;
  uint4 r;
  if (pmin + sx >= 0 && pmax + sx <= W - 1) {
    const int b0 = o0 + sx * C;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(srow + (b0 & ~3));
    const uint32_t sh = (uint32_t)(b0 & 3);
    uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    r.x = __builtin_amdgcn_alignbyte(w1, w0, sh);
    r.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
    r.z = __builtin_amdgcn_alignbyte(w3, w2, sh);
    r.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
  } else {
    uint32_t words[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        int o = o0 + j * 4 + b;
        int p = o / C, ch = o - p * C;
        int sp = min(max(p + sx, 0), W - 1);
        v |= (uint32_t)srow[sp * C + ch] << (8 * b);
      }
      words[j] = v;
    }
    r = make_uint4(words[0], words[1], words[2], words[3]);
  }
  return r;
}

// record gather: one thread per (sample, float of the record)
__device__ __forceinline__ void gather_record(const GatherArgs& a, int e) {
  const int i = e / a.rec_len, f = e - i * a.rec_len;
  if (i >= a.batch) return;
  const int buf = (i < a.count0) ? 0 : 1;
  const int64_t slot = a.idx[buf][buf == 0 ? i : i - a.count0];
  const float v = a.rec[buf][(size_t)slot * a.rec_len + f];
  const int S = a.S, A = a.A;
  if (f < S) a.out_state[(size_t)i * S + f] = v;
  else if (f < 2 * S) a.out_state[((size_t)a.batch + i) * S + (f - S)] = v;
  else if (f < 2 * S + A) a.out_action[(size_t)i * A + (f - 2 * S)] = v;
  else if (f == 2 * S + A) a.out_reward[i] = v;
  else if (f == 2 * S + A + 1) a.out_mask[i] = v;
  else a.out_done[i] = (uint8_t)(v != 0.0f);
}

template <int CT>
__global__ __launch_bounds__(256) void gather_crop_kernel(GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int rowb = a.W * a.C;        // bytes per row
  const int lds_stride = rowb + 16;  // 16B pad: the 5th dword of the last chunk stays in bounds
  if ((int)blockIdx.x < a.n_frame_blocks) {
    const int chunks = (a.H + kRowsPerBlock - 1) / kRowsPerBlock;
    int bid = blockIdx.x;
    const int rc = bid % chunks;
    bid /= chunks;
    const int i = bid % a.batch;
    bid /= a.batch;
    const int cam = bid % a.n_cam;
    const int which = bid / a.n_cam;  // 0 = observation frame (slot idx-1), 1 = next frame (slot idx)
    const int buf = (i < a.count0) ? 0 : 1;
    const size_t fbytes = (size_t)a.H * rowb;
    const uint8_t* src;
    if (a.from_packed) {
      src = a.packed[cam] + ((size_t)i * 2 + which) * fbytes;
    } else {
      // window start = idx - T; numpy wraps a negative window index to cap - T + (idx - T)
      // (reference quirk for a valid slot 0, see oracle/replay_oracle.py gather()).  T == 1 here.
      int64_t start = a.idx[buf][buf == 0 ? i : i - a.count0] - 1;
      if (start < 0) start += a.cap[buf] - 1;
      src = a.frames[buf][cam] + (size_t)(start + which) * fbytes;
    }
    const int32_t* crop = which == 0 ? a.crop_obs : a.crop_next;
    const int dy = crop ? crop[2 * i] : 4, dx = crop ? crop[2 * i + 1] : 4;
    const int sy = dy - 4, sx = dx - 4;
    const int h0 = rc * kRowsPerBlock;
    const int nrows = min(kRowsPerBlock, a.H - h0);
    const int vec_per_row = rowb / 16;
    // stage: LDS row r <- source row clamp(h0 + r + sy)
    for (int v = tid; v < nrows * vec_per_row; v += 256) {
      const int r = v / vec_per_row, q = v - r * vec_per_row;
      const int sh = min(max(h0 + r + sy, 0), a.H - 1);
      const uint4 val = *reinterpret_cast<const uint4*>(src + (size_t)sh * rowb + q * 16);
      *reinterpret_cast<uint4*>(lds + r * lds_stride + q * 16) = val;
    }
    __syncthreads();
    uint8_t* dst = a.out_frames + (((size_t)which * a.n_cam + cam) * a.batch + i) * fbytes +
                   (size_t)h0 * rowb;
    for (int v = tid; v < nrows * vec_per_row; v += 256) {
      const int r = v / vec_per_row, q = v - r * vec_per_row;
      const uint4 val = shifted_chunk<CT>(lds + r * lds_stride, q, sx, a.W, a.C);
      *reinterpret_cast<uint4*>(dst + (size_t)r * rowb + q * 16) = val;
    }
  } else if (!a.from_packed) {
    gather_record(a, (blockIdx.x - a.n_frame_blocks) * 256 + tid);
  }
}

// RGB frames (C == 3, W*3 a multiple of 16 and >= 32): no LDS.  Each output 16-byte vector is ONE unaligned 16-byte
// global load at byte offset q*16 + 3*sx of the (clamped) source row -- gfx950 global loads take any byte address --
// so a workgroup is a pure stream of independent load -> store pairs (4 vectors per thread, all loads in flight
// before the first store), with no barrier between a staging and a shifting phase.  Only the first / last vector of
// a row can reach past the row when sx != 0: its load is clamped into the row and the result is shifted by n = 3|sx|
// bytes with the border pixel replicated into the vacated bytes (n is a multiple of 3, so the replicated pattern's
// phase does not depend on n).  The arithmetic is uniform: no divergent edge path.
struct __attribute__((packed, aligned(1))) U128Unaligned { unsigned __int128 v; };
constexpr int kDirectVec = 4;  // vectors per thread

__global__ __launch_bounds__(256) void gather_crop_rgb_kernel(GatherArgs a) {
  const int tid = threadIdx.x;
  const int rowb = a.W * 3;
  const int vec_per_row = rowb >> 4;
  const int nvec = a.H * vec_per_row;
  if ((int)blockIdx.x < a.n_frame_blocks) {
    const int parts = (nvec + 256 * kDirectVec - 1) / (256 * kDirectVec);
    int bid = blockIdx.x;
    const int part = bid % parts;
    bid /= parts;
    const int i = bid % a.batch;
    bid /= a.batch;
    const int cam = bid % a.n_cam;
    const int which = bid / a.n_cam;
    const int buf = (i < a.count0) ? 0 : 1;
    const size_t fbytes = (size_t)a.H * rowb;
    const uint8_t* src;
    if (a.from_packed) {
      src = a.packed[cam] + ((size_t)i * 2 + which) * fbytes;
    } else {
      int64_t start = a.idx[buf][buf == 0 ? i : i - a.count0] - 1;
      if (start < 0) start += a.cap[buf] - 1;  // numpy negative window index (reference quirk, see gather_crop_kernel)
      src = a.frames[buf][cam] + (size_t)(start + which) * fbytes;
    }
    const int32_t* crop = which == 0 ? a.crop_obs : a.crop_next;
    const int dy = crop ? crop[2 * i] : 4, dx = crop ? crop[2 * i + 1] : 4;
    const int sy = dy - 4, sx3 = (dx - 4) * 3;
    uint8_t* dst = a.out_frames + (((size_t)which * a.n_cam + cam) * a.batch + i) * fbytes;
    unsigned __int128 val[kDirectVec];
    int shl[kDirectVec], shr[kDirectVec];
#pragma unroll
    for (int j = 0; j < kDirectVec; ++j) {
      const int v = min((part * kDirectVec + j) * 256 + tid, nvec - 1);
      const int r = v / vec_per_row, q = v - r * vec_per_row;
      const int sh = min(max(r + sy, 0), a.H - 1);
      const int b0 = q * 16 + sx3;
      const int bc = min(max(b0, 0), rowb - 16);
      shl[j] = (bc - b0) * 8;   // > 0: left border, vector starts before the row
      shr[j] = (b0 - bc) * 8;   // > 0: right border
      val[j] = reinterpret_cast<const U128Unaligned*>(src + (size_t)sh * rowb + bc)->v;
    }
#pragma unroll
    for (int j = 0; j < kDirectVec; ++j) {
      const int v = (part * kDirectVec + j) * 256 + tid;
      unsigned __int128 x = val[j];
      const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 96);
      if (shl[j] > 0) {
        // bytes p0 p1 p2 of the first pixel; pattern byte k = p[k % 3]
        const uint32_t d0 = __builtin_amdgcn_perm(0, lo, 0x00020100u);  // p0 p1 p2 p0
        const uint32_t d1 = __builtin_amdgcn_perm(0, lo, 0x01000201u);  // p1 p2 p0 p1
        const uint32_t d2 = __builtin_amdgcn_perm(0, lo, 0x02010002u);  // p2 p0 p1 p2
        const unsigned __int128 pat = (unsigned __int128)d0 | ((unsigned __int128)d1 << 32) | ((unsigned __int128)d2 << 64) |
                                      ((unsigned __int128)d0 << 96);
        const unsigned __int128 keep = (~(unsigned __int128)0) << shl[j];
        x = (x << shl[j]) | (pat & ~keep);
      } else if (shr[j] > 0) {
        // bytes l0 l1 l2 of the last pixel = bytes 13..15 of the vector; pattern byte k = l[(k + 2) % 3]
        const uint32_t d0 = __builtin_amdgcn_perm(0, hi, 0x03020103u);  // l2 l0 l1 l2   (hi bytes: 1,2,3 = l0,l1,l2)
        const uint32_t d1 = __builtin_amdgcn_perm(0, hi, 0x01030201u);  // l0 l1 l2 l0
        const uint32_t d2 = __builtin_amdgcn_perm(0, hi, 0x02010302u);  // l1 l2 l0 l1
        const unsigned __int128 pat = (unsigned __int128)d0 | ((unsigned __int128)d1 << 32) | ((unsigned __int128)d2 << 64) |
                                      ((unsigned __int128)d0 << 96);
        const unsigned __int128 keep = (~(unsigned __int128)0) >> shr[j];
        x = (x >> shr[j]) | (pat & ~keep);
      }
      if (v < nvec) {
        uint4 o;
        o.x = (uint32_t)x; o.y = (uint32_t)(x >> 32); o.z = (uint32_t)(x >> 64); o.w = (uint32_t)(x >> 96);
        *reinterpret_cast<uint4*>(dst + (size_t)v * 16) = o;
      }
    }
  } else if (!a.from_packed) {
    gather_record(a, (blockIdx.x - a.n_frame_blocks) * 256 + tid);
  }
}

struct PackedArgs {
  const uint8_t* frames[SERL_MAX_CAMS];
  const float* rec;
  const int64_t* idx;
  int batch, n_cam, T, TS, A, rec_len;
  int64_t cap;
  size_t fbytes;
  uint8_t* out_frames[SERL_MAX_CAMS];  // [batch][T+1][fbytes]
  float *out_state, *out_next_state, *out_action, *out_reward, *out_mask;
  uint8_t* out_done;
  int n_frame_blocks, vec_per_block;
};

// sample(pack_obs_and_next_obs=True): straight 16B-vector copy of slots idx-T..idx per camera
__global__ __launch_bounds__(256) void gather_packed_kernel(PackedArgs a) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.n_frame_blocks) {
    const int blocks_per_frame = (int)((a.fbytes / 16 + a.vec_per_block - 1) / a.vec_per_block);
    int bid = blockIdx.x;
    const int part = bid % blocks_per_frame;
    bid /= blocks_per_frame;
    const int t = bid % (a.T + 1);
    bid /= (a.T + 1);
    const int i = bid % a.batch;
    const int cam = bid / a.batch;
    int64_t start = a.idx[i] - a.T;
    if (start < 0) start += a.cap - a.T;  // numpy negative window index (reference quirk)
    const int64_t slot = start + t;
    const uint4* src = reinterpret_cast<const uint4*>(a.frames[cam] + (size_t)slot * a.fbytes);
    uint4* dst = reinterpret_cast<uint4*>(a.out_frames[cam] + ((size_t)i * (a.T + 1) + t) * a.fbytes);
    const int nvec = (int)(a.fbytes / 16);
    const int v0 = part * a.vec_per_block;
    for (int v = v0 + tid; v < min(v0 + a.vec_per_block, nvec); v += 256) dst[v] = src[v];
  } else {
    const int e = (blockIdx.x - a.n_frame_blocks) * 256 + tid;
    const int i = e / a.rec_len, f = e - i * a.rec_len;
    if (i >= a.batch) return;
    const float v = a.rec[(size_t)a.idx[i] * a.rec_len + f];
    const int S = a.TS, A = a.A;
    if (f < S) a.out_state[(size_t)i * S + f] = v;
    else if (f < 2 * S) a.out_next_state[(size_t)i * S + (f - S)] = v;
    else if (f < 2 * S + A) a.out_action[(size_t)i * A + (f - 2 * S)] = v;
    else if (f == 2 * S + A) a.out_reward[i] = v;
    else if (f == 2 * S + A + 1) a.out_mask[i] = v;
    else a.out_done[i] = (uint8_t)(v != 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------
// An insert must not overwrite a slot that an in-flight gather may still read: the copy stream waits (on the device)
// for the last enqueued gather.  Caller holds rb->mu.
static int order_after_gathers(serl_rb* rb) {
  if (rb->gather_pending) {
    for (int k = 0; k < serl_rb::kGatherStreams; ++k)
      if (rb->gather_pend[k]) {
        SERL_HIP(hipStreamWaitEvent(rb->copy_stream, rb->gather_ev[k], 0));
        rb->gather_pend[k] = false;
      }
    rb->gather_pending = false;
  }
  return SERL_OK;
}
// records "a gather of this buffer was enqueued on `stream`".  Caller holds rb->mu.
static int note_gather(serl_rb* rb, hipStream_t stream) {
  int k = -1;
  for (int i = 0; i < serl_rb::kGatherStreams && k < 0; ++i)
    if (rb->gather_used[i] && rb->gather_stream[i] == stream) k = i;
  if (k < 0) {   // a stream not seen before: take an entry with nothing pending, else recycle round-robin (the copy stream
                 // first waits for the recycled entry's gather, so nothing is forgotten)
    for (int i = 0; i < serl_rb::kGatherStreams && k < 0; ++i)
      if (!rb->gather_pend[i]) k = i;
    if (k < 0) {
      k = rb->gather_rr = (rb->gather_rr + 1) % serl_rb::kGatherStreams;
      SERL_HIP(hipStreamWaitEvent(rb->copy_stream, rb->gather_ev[k], 0));
    }
    rb->gather_stream[k] = stream;
    rb->gather_used[k] = true;
  }
  SERL_HIP(hipEventRecord(rb->gather_ev[k], stream));
  rb->gather_pend[k] = true;
  rb->gather_pending = true;
  return SERL_OK;
}
// ... and a gather enqueued after an insert sees it: `stream` waits for the last insert's copies.  Caller holds rb->mu.
// Once the last insert's event has COMPLETED the flag is cleared and later gathers carry no wait at all: until round 5 the flag
// was never cleared, so every gather -- the first command of every trunk pass -- opened with a cross-stream wait on a
// long-finished event (same-call A/B: pipelined 2.4497 / 2.4696 -> 2.4431 / 2.4544 ms, serial unchanged; NOT the cause of the
// 80-110 us of idle time at the pass boundary, profiles/README.md).
static int order_after_inserts(serl_rb* rb, hipStream_t stream) {
  if (!rb->insert_pending) return SERL_OK;
  const hipError_t q = hipEventQuery(rb->last_insert);
  if (q == hipSuccess) { rb->insert_pending = false; return SERL_OK; }
  if (q != hipErrorNotReady) SERL_HIP(q);
  (void)hipGetLastError();   // (hipErrorNotReady is sticky in hipGetLastError)
  SERL_HIP(hipStreamWaitEvent(stream, rb->last_insert, 0));
  return SERL_OK;
}

// writes slot `i` (record + one frame per camera) host -> HBM through the pinned ring.  Caller holds rb->mu.
static int write_slot(serl_rb* rb, int64_t i, const uint8_t* const* frames_host, const float* rec) {
  std::memcpy(&rb->rec_host[(size_t)i * rb->rec_len], rec, sizeof(float) * rb->rec_len);
  const int s = rb->ins_next;
  rb->ins_next = (s + 1) % kInsRing;
  if (rb->ins_used[s]) SERL_HIP(hipEventSynchronize(rb->ins_done[s]));  // ring wrapped: that copy is long done
  uint8_t* h = rb->ins_host + (size_t)s * rb->ins_slot_bytes;
  const size_t rec_bytes = sizeof(float) * rb->rec_len;
  std::memcpy(h, rec, rec_bytes);
  const size_t f0 = (rec_bytes + 255) & ~(size_t)255;
  for (int c = 0; c < rb->n_cam; ++c) std::memcpy(h + f0 + (size_t)c * rb->frame_bytes, frames_host[c], rb->frame_bytes);
  SERL_HIP(hipMemcpyAsync(rb->rec + (size_t)i * rb->rec_len, h, rec_bytes, hipMemcpyHostToDevice, rb->copy_stream));
  for (int c = 0; c < rb->n_cam; ++c)
    SERL_HIP(hipMemcpyAsync(rb->frames[c] + (size_t)i * rb->frame_bytes, h + f0 + (size_t)c * rb->frame_bytes,
                            rb->frame_bytes, hipMemcpyHostToDevice, rb->copy_stream));
  SERL_HIP(hipEventRecord(rb->ins_done[s], rb->copy_stream));
  rb->ins_used[s] = true;
  rb->insert_index = (i + 1) % rb->cap;
  rb->size = rb->size + 1 < rb->cap ? rb->size + 1 : rb->cap;
  return SERL_OK;
}

// device->device copy of slot src to the write head (wrap re-insert,
// memory_efficient_replay_buffer.py:54-59).  Caller holds rb->mu.
static int copy_slot_to_head(serl_rb* rb, int64_t src) {
  const int64_t i = rb->insert_index;
  std::memmove(&rb->rec_host[(size_t)i * rb->rec_len], &rb->rec_host[(size_t)src * rb->rec_len],
               sizeof(float) * rb->rec_len);
  SERL_HIP(hipMemcpyAsync(rb->rec + (size_t)i * rb->rec_len, rb->rec + (size_t)src * rb->rec_len,
                          sizeof(float) * rb->rec_len, hipMemcpyDeviceToDevice, rb->copy_stream));
  for (int c = 0; c < rb->n_cam; ++c)
    SERL_HIP(hipMemcpyAsync(rb->frames[c] + (size_t)i * rb->frame_bytes,
                            rb->frames[c] + (size_t)src * rb->frame_bytes, rb->frame_bytes,
                            hipMemcpyDeviceToDevice, rb->copy_stream));
  rb->insert_index = (i + 1) % rb->cap;
  rb->size = rb->size + 1 < rb->cap ? rb->size + 1 : rb->cap;
  return SERL_OK;
}

// end of an insert: later gathers wait for its copies.  Caller holds rb->mu.
static int finish_insert(serl_rb* rb) {
  SERL_HIP(hipEventRecord(rb->last_insert, rb->copy_stream));
  rb->insert_pending = true;
  return SERL_OK;
}

// reserve a staging slot, copy `bytes` of host parameters into pinned memory and enqueue the H2D
// copy on `stream`.  Returns the device address of the slot.
static int stage_params(serl_rb* rb, const void* const* srcs, const size_t* sizes, int n,
                        hipStream_t stream, uint8_t** dev_out, size_t* offsets) {
  const int s = rb->stage_next;
  rb->stage_next = (s + 1) % kRing;
  if (rb->stage_used[s]) SERL_HIP(hipEventSynchronize(rb->stage_done[s]));
  uint8_t* h = rb->stage_host + (size_t)s * rb->stage_slot_bytes;
  size_t off = 0;
  for (int k = 0; k < n; ++k) {
    offsets[k] = off;
    if (srcs[k]) std::memcpy(h + off, srcs[k], sizes[k]);
    off += (sizes[k] + 15) & ~(size_t)15;
    if (off > rb->stage_slot_bytes) {
      set_error("batch too large for the staging slot (%zu > %zu bytes)", off, rb->stage_slot_bytes);
      return SERL_ERR_INVALID;
    }
  }
  uint8_t* d = rb->stage_dev + (size_t)s * rb->stage_slot_bytes;
  // (round 5: letting the gather kernel read the pinned host slot itself -- no copy command on the stream -- left the step
  //  unchanged, 2.5225 / 2.5198 -> 2.5221 / 2.5239 ms: the idle time in front of a pass is not the copy's, profiles/README.md)
  SERL_HIP(hipMemcpyAsync(d, h, off, hipMemcpyHostToDevice, stream));
  *dev_out = d;
  rb->stage_used[s] = true;
  return s;
}

}  // namespace serl

using namespace serl;

extern "C" {

const char* serl_last_error(void) { return serl::g_err; }
int serl_version(void) { return 100; }
int serl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int serl_rb_create(int device, int64_t capacity, int n_cam, int H, int W, int C, int num_stack,
                   int state_dim, int act_dim, serl_rb** out) {
  SERL_REQUIRE(out != nullptr, "out is NULL");
  SERL_REQUIRE(capacity > 1 && capacity < 0x7FFFFFFF, "capacity %lld out of range", (long long)capacity);
  // n_cam == 0: plain ReplayBuffer of flat observations (replay_buffer.py:41-75): every inserted slot is valid
  SERL_REQUIRE(n_cam >= 0 && n_cam <= SERL_MAX_CAMS, "n_cam %d not in [0,%d]", n_cam, SERL_MAX_CAMS);
  SERL_REQUIRE(num_stack >= 1, "num_stack must be >= 1");
  if (n_cam == 0) { H = W = C = 0; }
  SERL_REQUIRE(n_cam == 0 || ((size_t)W * C) % 16 == 0, "W*C (%d) must be a multiple of 16 bytes", W * C);
  SERL_REQUIRE((n_cam == 0 || H >= 1) && state_dim >= 1 && act_dim >= 1, "bad dims");
  SERL_HIP(hipSetDevice(device));
  serl_rb* rb = new serl_rb();
  rb->device = device;
  rb->cap = capacity;
  rb->n_cam = n_cam;
  rb->H = H; rb->W = W; rb->C = C; rb->T = num_stack; rb->S = state_dim; rb->A = act_dim;
  rb->rec_len = 2 * num_stack * state_dim + act_dim + 3;
  rb->frame_bytes = (size_t)H * W * C;
  rb->valid.assign((size_t)capacity, 0);
  rb->rec_host.assign((size_t)capacity * rb->rec_len, 0.0f);
  for (int c = 0; c < n_cam; ++c) {
    hipError_t e = hipMalloc((void**)&rb->frames[c], (size_t)capacity * rb->frame_bytes);
    if (e != hipSuccess) {
      set_error("hipMalloc of %zu bytes for camera %d failed: %s", (size_t)capacity * rb->frame_bytes,
                c, hipGetErrorString(e));
      serl_rb_destroy(rb);
      return SERL_ERR_HIP;
    }
  }
  SERL_HIP(hipMalloc((void**)&rb->rec, (size_t)capacity * rb->rec_len * sizeof(float)));
  SERL_HIP(hipStreamCreateWithFlags(&rb->copy_stream, hipStreamNonBlocking));
  for (int k = 0; k < serl_rb::kGatherStreams; ++k) SERL_HIP(hipEventCreateWithFlags(&rb->gather_ev[k], hipEventDisableTiming));
  SERL_HIP(hipEventCreateWithFlags(&rb->last_insert, hipEventDisableTiming));
  rb->ins_slot_bytes = ((sizeof(float) * rb->rec_len + 255) & ~(size_t)255) + (size_t)n_cam * rb->frame_bytes;
  SERL_HIP(hipHostMalloc((void**)&rb->ins_host, rb->ins_slot_bytes * kInsRing, hipHostMallocDefault));
  for (int s = 0; s < kInsRing; ++s) SERL_HIP(hipEventCreateWithFlags(&rb->ins_done[s], hipEventDisableTiming));
  rb->stage_slot_bytes = 1 << 16;  // idx (8B) + 2 crops (16B) per sample: up to ~2700 samples
  SERL_HIP(hipHostMalloc((void**)&rb->stage_host, rb->stage_slot_bytes * kRing, hipHostMallocDefault));
  SERL_HIP(hipMalloc((void**)&rb->stage_dev, rb->stage_slot_bytes * kRing));
  for (int s = 0; s < kRing; ++s)
    SERL_HIP(hipEventCreateWithFlags(&rb->stage_done[s], hipEventDisableTiming));
  *out = rb;
  return SERL_OK;
}

int serl_rb_destroy(serl_rb* rb) {
  if (!rb) return SERL_OK;
  (void)hipSetDevice(rb->device);
  if (rb->copy_stream) (void)hipStreamSynchronize(rb->copy_stream);
  for (int c = 0; c < SERL_MAX_CAMS; ++c)
    if (rb->frames[c]) (void)hipFree(rb->frames[c]);
  if (rb->rec) (void)hipFree(rb->rec);
  if (rb->stage_host) (void)hipHostFree(rb->stage_host);
  if (rb->stage_dev) (void)hipFree(rb->stage_dev);
  for (int s = 0; s < kRing; ++s)
    if (rb->stage_done[s]) (void)hipEventDestroy(rb->stage_done[s]);
  for (int k = 0; k < serl_rb::kGatherStreams; ++k)
    if (rb->gather_ev[k]) (void)hipEventDestroy(rb->gather_ev[k]);
  if (rb->last_insert) (void)hipEventDestroy(rb->last_insert);
  if (rb->ins_host) (void)hipHostFree(rb->ins_host);
  for (int s = 0; s < kInsRing; ++s)
    if (rb->ins_done[s]) (void)hipEventDestroy(rb->ins_done[s]);
  if (rb->copy_stream) (void)hipStreamDestroy(rb->copy_stream);
  delete rb;
  return SERL_OK;
}

int serl_rb_seed(serl_rb* rb, uint64_t state_hi, uint64_t state_lo, uint64_t inc_hi,
                 uint64_t inc_lo, int has_uint32, uint32_t uinteger) {
  SERL_REQUIRE(rb, "rb is NULL");
  std::lock_guard<std::mutex> g(rb->mu);
  rb->rng.state = ((unsigned __int128)state_hi << 64) | state_lo;
  rb->rng.inc = ((unsigned __int128)inc_hi << 64) | inc_lo;
  rb->rng.has_uint32 = has_uint32;
  rb->rng.uinteger = uinteger;
  rb->rng.seeded = true;
  return SERL_OK;
}

int serl_rb_rng_state(serl_rb* rb, uint64_t out[4], int* has_uint32, uint32_t* uinteger) {
  SERL_REQUIRE(rb && out && has_uint32 && uinteger, "NULL argument");
  std::lock_guard<std::mutex> g(rb->mu);
  out[0] = (uint64_t)(rb->rng.state >> 64);
  out[1] = (uint64_t)rb->rng.state;
  out[2] = (uint64_t)(rb->rng.inc >> 64);
  out[3] = (uint64_t)rb->rng.inc;
  *has_uint32 = rb->rng.has_uint32;
  *uinteger = rb->rng.uinteger;
  return SERL_OK;
}

int serl_rb_insert(serl_rb* rb, const uint8_t* const* obs_frames, const uint8_t* const* next_frames,
                   const float* state, const float* next_state, const float* action, float reward,
                   float mask, int done) {
  SERL_REQUIRE(rb && state && next_state && action, "NULL argument");
  SERL_REQUIRE(rb->n_cam == 0 || (obs_frames && next_frames), "NULL frames");
  std::lock_guard<std::mutex> g(rb->mu);
  SERL_HIP(hipSetDevice(rb->device));
  int rc = order_after_gathers(rb);  // never overwrite a slot an in-flight gather may still read
  if (rc) return rc;
  const int T = rb->T, TS = rb->T * rb->S;
  if (rb->n_cam == 0) {  // ReplayBuffer.insert (replay_buffer.py:71-75): write at the head, advance, no bookkeeping
    std::vector<float> r0(rb->rec_len);
    std::memcpy(r0.data(), state, sizeof(float) * TS);
    std::memcpy(r0.data() + TS, next_state, sizeof(float) * TS);
    std::memcpy(r0.data() + 2 * TS, action, sizeof(float) * rb->A);
    r0[2 * TS + rb->A] = reward;
    r0[2 * TS + rb->A + 1] = mask;
    r0[2 * TS + rb->A + 2] = done ? 1.0f : 0.0f;
    rb->valid[rb->insert_index] = 1;
    if ((rc = write_slot(rb, rb->insert_index, nullptr, r0.data()))) return rc;
    return finish_insert(rb);
  }
  // wrap: re-insert the last T slots at the head as invalid (py:54-59)
  if (rb->insert_index == 0 && rb->cap == rb->size && !rb->first) {
    for (int64_t src = rb->size - T; src < rb->size; ++src) {
      rb->valid[rb->insert_index] = 0;
      if ((rc = copy_slot_to_head(rb, src))) return rc;
    }
  }
  std::vector<float> rec(rb->rec_len);
  std::memcpy(rec.data(), state, sizeof(float) * TS);
  std::memcpy(rec.data() + TS, next_state, sizeof(float) * TS);
  std::memcpy(rec.data() + 2 * TS, action, sizeof(float) * rb->A);
  rec[2 * TS + rb->A] = reward;
  rec[2 * TS + rb->A + 1] = mask;
  rec[2 * TS + rb->A + 2] = done ? 1.0f : 0.0f;
  const uint8_t* fr[SERL_MAX_CAMS];
  if (rb->first) {  // episode start: T invalid "first-frame" slots holding the obs frames (py:71-77)
    for (int t = 0; t < T; ++t) {
      for (int c = 0; c < rb->n_cam; ++c) fr[c] = obs_frames[c] + (size_t)t * rb->frame_bytes;
      rb->valid[rb->insert_index] = 0;
      if ((rc = write_slot(rb, rb->insert_index, fr, rec.data()))) return rc;
    }
  }
  for (int c = 0; c < rb->n_cam; ++c) fr[c] = next_frames[c] + (size_t)(T - 1) * rb->frame_bytes;
  rb->first = done != 0;
  rb->valid[rb->insert_index] = 1;
  if ((rc = write_slot(rb, rb->insert_index, fr, rec.data()))) return rc;
  for (int t = 0; t < T; ++t) rb->valid[(rb->insert_index + t) % rb->size] = 0;  // py:87-89
  return finish_insert(rb);
}

int64_t serl_rb_len(serl_rb* rb) {
  if (!rb) return -1;
  std::lock_guard<std::mutex> g(rb->mu);
  return rb->size;
}
int64_t serl_rb_insert_index(serl_rb* rb) {
  if (!rb) return -1;
  std::lock_guard<std::mutex> g(rb->mu);
  return rb->insert_index;
}
int serl_rb_valid_mask(serl_rb* rb, uint8_t* host_out) {
  SERL_REQUIRE(rb && host_out, "NULL argument");
  std::lock_guard<std::mutex> g(rb->mu);
  std::memcpy(host_out, rb->valid.data(), (size_t)rb->cap);
  return SERL_OK;
}

int serl_rb_sample_indices(serl_rb* rb, int batch, int64_t* host_idx_out) {
  SERL_REQUIRE(rb && host_idx_out, "NULL argument");
  SERL_REQUIRE(batch >= 0, "negative batch");
  std::lock_guard<std::mutex> g(rb->mu);
  if (!rb->rng.seeded) {
    set_error("replay buffer RNG not seeded: call serl_rb_seed first");
    return SERL_ERR_STATE;
  }
  if (batch > 0 && rb->size <= 0) {
    set_error("cannot sample from an empty replay buffer");
    return SERL_ERR_STATE;
  }
  const uint32_t n = (uint32_t)rb->size;
  if (batch > 0) {
    bool any = false;
    for (int64_t i = 0; i < rb->size && !any; ++i) any = rb->valid[i];
    if (!any) {
      set_error("replay buffer holds no valid transition");
      return SERL_ERR_STATE;
    }
  }
  for (int i = 0; i < batch; ++i) host_idx_out[i] = rb->rng.bounded(n);  // integers(len, size=B)
  for (int i = 0; i < batch; ++i)
    while (!rb->valid[host_idx_out[i]]) host_idx_out[i] = rb->rng.bounded(n);  // rejection loop
  return SERL_OK;
}

// The reference holds one lock across index draw and gather (data_store.py:108-111); here they are two calls (the
// lazy / prefetched path), so an insert in between may have invalidated a drawn slot (the look-ahead invalidation of
// memory_efficient_replay_buffer.py:87-89, a new episode's first-frame slot, the wrap re-insert).  A slot that is still
// valid pairs with slot-1 consistently (writes are sequential), so validity is the whole check: stale indices are
// re-drawn from the buffer's generator under the lock, exactly as the rejection loop would have.  `out` stays empty
// when nothing changed.  Caller holds rb->mu.
// An index drawn before an insert invalidated its slot is re-drawn here, IN PLACE: the caller's array then describes the
// batch that is actually gathered (index-keyed bookkeeping and determinism checks stay valid).
static int revalidate(serl_rb* rb, int64_t* idx, int n) {
  if (rb->n_cam == 0) return SERL_OK;   // plain ReplayBuffer: every slot below `size` is valid
  for (int i = 0; i < n; ++i) {
    if (rb->valid[idx[i]]) continue;
    SERL_REQUIRE(rb->rng.seeded, "replay buffer RNG not seeded");
    const uint32_t sz = (uint32_t)rb->size;
    int guard = 0;
    do {
      idx[i] = rb->rng.bounded(sz);
      SERL_REQUIRE(++guard < (1 << 24), "no valid slot found while re-drawing a stale index");
    } while (!rb->valid[idx[i]]);
  }
  return SERL_OK;
}

static int check_indices(serl_rb* rb, const int64_t* idx, int n) {
  for (int i = 0; i < n; ++i) {
    if (idx[i] < 0 || idx[i] >= rb->size) {
      set_error("index %lld out of range [0,%lld)", (long long)idx[i], (long long)rb->size);
      return SERL_ERR_INVALID;
    }
  }
  return SERL_OK;
}

int serl_rb_gather_packed(serl_rb* rb, int64_t* host_idx, int batch,
                          uint8_t* const* dev_frames_out, float* dev_state_out,
                          float* dev_next_state_out, float* dev_action_out, float* dev_reward_out,
                          float* dev_mask_out, uint8_t* dev_done_out, void* stream_) {
  SERL_REQUIRE(rb && host_idx && (dev_frames_out || rb->n_cam == 0), "NULL argument");
  SERL_REQUIRE(batch > 0, "batch must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  std::lock_guard<std::mutex> g(rb->mu);
  SERL_HIP(hipSetDevice(rb->device));
  int rc = check_indices(rb, host_idx, batch);
  if (rc) return rc;
  if ((rc = revalidate(rb, host_idx, batch))) return rc;
  if ((rc = order_after_inserts(rb, stream))) return rc;
  const void* srcs[1] = {host_idx};
  size_t sizes[1] = {sizeof(int64_t) * (size_t)batch}, offs[1];
  uint8_t* dparams;
  int slot = stage_params(rb, srcs, sizes, 1, stream, &dparams, offs);
  if (slot < 0) return slot;
  PackedArgs a{};
  for (int c = 0; c < rb->n_cam; ++c) {
    a.frames[c] = rb->frames[c];
    a.out_frames[c] = dev_frames_out[c];
  }
  a.rec = rb->rec;
  a.idx = reinterpret_cast<const int64_t*>(dparams + offs[0]);
  a.batch = batch; a.n_cam = rb->n_cam; a.T = rb->T; a.TS = rb->T * rb->S; a.A = rb->A;
  a.rec_len = rb->rec_len; a.fbytes = rb->frame_bytes; a.cap = rb->cap;
  a.out_state = dev_state_out; a.out_next_state = dev_next_state_out; a.out_action = dev_action_out;
  a.out_reward = dev_reward_out; a.out_mask = dev_mask_out; a.out_done = dev_done_out;
  a.vec_per_block = 256 * 4;
  const int blocks_per_frame = cdiv((long)(rb->frame_bytes / 16), a.vec_per_block);
  a.n_frame_blocks = rb->n_cam * batch * (rb->T + 1) * blocks_per_frame;
  const int rec_blocks = dev_state_out ? cdiv((long)batch * rb->rec_len, 256) : 0;
  hipLaunchKernelGGL(gather_packed_kernel, dim3(a.n_frame_blocks + rec_blocks), dim3(256), 0, stream, a);
  SERL_HIP(hipGetLastError());
  SERL_HIP(hipEventRecord(rb->stage_done[slot], stream));
  return note_gather(rb, stream);
}

static int launch_gather_crop(GatherArgs& a, hipStream_t stream) {
  const int rec_blocks = a.from_packed ? 0 : cdiv((long)a.batch * a.rec_len, 256);
  ProfScope prof("gather_crop", stream);
  if (a.C == 3 && (a.W * 3) % 16 == 0 && a.W * 3 >= 32) {   // RGB rows of whole 16-byte vectors: the LDS-free kernel
    const int nvec = a.H * (a.W * 3 / 16);
    a.n_frame_blocks = 2 * a.n_cam * a.batch * cdiv(nvec, 256 * kDirectVec);
    hipLaunchKernelGGL(gather_crop_rgb_kernel, dim3(a.n_frame_blocks + rec_blocks), dim3(256), 0, stream, a);
    SERL_HIP(hipGetLastError());
    return SERL_OK;
  }
  const int chunks = cdiv(a.H, kRowsPerBlock);
  a.n_frame_blocks = 2 * a.n_cam * a.batch * chunks;
  const size_t lds = (size_t)kRowsPerBlock * ((size_t)a.W * a.C + 16);
  if (a.C == 3) hipLaunchKernelGGL(gather_crop_kernel<3>, dim3(a.n_frame_blocks + rec_blocks), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(gather_crop_kernel<0>, dim3(a.n_frame_blocks + rec_blocks), dim3(256), lds, stream, a);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

int serl_rb_gather_crop(serl_rb* const* rbs, int n_rb, int64_t* const* host_idx,
                        const int* counts, const int32_t* host_crop_obs,
                        const int32_t* host_crop_next, const serl_batch* out, void* stream_) {
  SERL_REQUIRE(rbs && host_idx && counts && out, "NULL argument");
  SERL_REQUIRE(n_rb >= 1 && n_rb <= SERL_MAX_BUFFERS, "n_rb %d not in [1,%d]", n_rb, SERL_MAX_BUFFERS);
  hipStream_t stream = (hipStream_t)stream_;
  serl_rb* r0 = rbs[0];
  SERL_REQUIRE(r0, "rbs[0] is NULL");
  SERL_REQUIRE(r0->T == 1, "fused gather+crop supports num_stack == 1 (got %d)", r0->T);
  int total = 0;
  for (int b = 0; b < n_rb; ++b) {
    SERL_REQUIRE(rbs[b] && host_idx[b], "NULL buffer/index");
    SERL_REQUIRE(counts[b] >= 0, "negative count");
    SERL_REQUIRE(rbs[b]->n_cam == r0->n_cam && rbs[b]->H == r0->H && rbs[b]->W == r0->W &&
                     rbs[b]->C == r0->C && rbs[b]->S == r0->S && rbs[b]->A == r0->A &&
                     rbs[b]->T == r0->T && rbs[b]->device == r0->device,
                 "buffers have different shapes");
    total += counts[b];
  }
  SERL_REQUIRE(total == out->batch && total > 0, "counts sum %d != batch %d", total, out->batch);
  SERL_REQUIRE(out->n_cam == r0->n_cam && (r0->n_cam == 0 || (out->H == r0->H && out->W == r0->W && out->C == r0->C)) &&
                   out->state_dim == r0->S && out->act_dim == r0->A, "serl_batch shape mismatch");
  SERL_REQUIRE((out->frames || r0->n_cam == 0) && out->state && out->action && out->reward && out->mask && out->done,
               "serl_batch has NULL outputs");
  for (int k = 0; k < 2; ++k) {
    const int32_t* cr = k ? host_crop_next : host_crop_obs;
    if (cr)
      for (int i = 0; i < 2 * total; ++i)
        SERL_REQUIRE(cr[i] >= 0 && cr[i] <= 8, "crop offset %d out of [0,8]", cr[i]);
  }
  // lock all buffers (fixed order) while we read bookkeeping and enqueue
  std::unique_lock<std::mutex> l0(rbs[0]->mu, std::defer_lock), l1;
  if (n_rb == 2 && rbs[1] != rbs[0]) {
    l1 = std::unique_lock<std::mutex>(rbs[1]->mu, std::defer_lock);
    std::lock(l0, l1);
  } else {
    l0.lock();
  }
  SERL_HIP(hipSetDevice(r0->device));
  const int64_t* use_idx[SERL_MAX_BUFFERS] = {nullptr};
  for (int b = 0; b < n_rb; ++b) {
    int rc = check_indices(rbs[b], host_idx[b], counts[b]);
    if (rc) return rc;
    if ((rc = revalidate(rbs[b], host_idx[b], counts[b]))) return rc;
    use_idx[b] = host_idx[b];
    if ((rc = order_after_inserts(rbs[b], stream))) return rc;
  }
  const void* srcs[4] = {use_idx[0], n_rb > 1 ? use_idx[1] : nullptr, host_crop_obs, host_crop_next};
  size_t sizes[4] = {sizeof(int64_t) * (size_t)counts[0],
                     n_rb > 1 ? sizeof(int64_t) * (size_t)counts[1] : 0,
                     host_crop_obs ? sizeof(int32_t) * 2 * (size_t)total : 0,
                     host_crop_next ? sizeof(int32_t) * 2 * (size_t)total : 0};
  size_t offs[4];
  uint8_t* dparams;
  int slot = stage_params(r0, srcs, sizes, 4, stream, &dparams, offs);
  if (slot < 0) return slot;
  GatherArgs a{};
  for (int b = 0; b < n_rb; ++b) {
    for (int c = 0; c < r0->n_cam; ++c) a.frames[b][c] = rbs[b]->frames[c];
    a.rec[b] = rbs[b]->rec;
    a.cap[b] = rbs[b]->cap;
    a.idx[b] = reinterpret_cast<const int64_t*>(dparams + offs[b]);
  }
  a.count0 = counts[0];
  a.batch = total; a.n_cam = r0->n_cam; a.H = r0->H; a.W = r0->W; a.C = r0->C; a.S = r0->S; a.A = r0->A;
  a.rec_len = r0->rec_len;
  a.crop_obs = host_crop_obs ? reinterpret_cast<const int32_t*>(dparams + offs[2]) : nullptr;
  a.crop_next = host_crop_next ? reinterpret_cast<const int32_t*>(dparams + offs[3]) : nullptr;
  a.out_frames = out->frames; a.out_state = out->state; a.out_action = out->action;
  a.out_reward = out->reward; a.out_mask = out->mask; a.out_done = out->done;
  a.from_packed = 0;
  int rc = launch_gather_crop(a, stream);
  if (rc) return rc;
  SERL_HIP(hipEventRecord(r0->stage_done[slot], stream));
  for (int b = 0; b < n_rb; ++b) {
    int rc2 = note_gather(rbs[b], stream);
    if (rc2) return rc2;
  }
  return SERL_OK;
}

// Standalone crop needs its own parameter staging (no buffer handle): a small static pool.
namespace {
struct CropStage {
  std::mutex mu;
  int device = -1;
  uint8_t* host = nullptr;
  uint8_t* dev = nullptr;
  hipEvent_t done[serl::kRing] = {nullptr};
  bool used[serl::kRing] = {false};
  int next = 0;
  size_t slot_bytes = 1 << 16;
};
CropStage g_crop;
}  // namespace

int serl_crop_packed(int device, const uint8_t* const* dev_packed, int n_cam, int batch, int H,
                     int W, int C, const int32_t* host_crop_obs, const int32_t* host_crop_next,
                     uint8_t* dev_frames_out, void* stream_) {
  SERL_REQUIRE(dev_packed && dev_frames_out, "NULL argument");
  SERL_REQUIRE(n_cam >= 1 && n_cam <= SERL_MAX_CAMS && batch > 0, "bad n_cam/batch");
  SERL_REQUIRE(((size_t)W * C) % 16 == 0, "W*C (%d) must be a multiple of 16 bytes", W * C);
  hipStream_t stream = (hipStream_t)stream_;
  std::lock_guard<std::mutex> g(g_crop.mu);
  SERL_HIP(hipSetDevice(device));
  if (g_crop.device != device) {
    SERL_REQUIRE(g_crop.device == -1, "serl_crop_packed is bound to device %d", g_crop.device);
    SERL_HIP(hipHostMalloc((void**)&g_crop.host, g_crop.slot_bytes * kRing, hipHostMallocDefault));
    SERL_HIP(hipMalloc((void**)&g_crop.dev, g_crop.slot_bytes * kRing));
    for (int s = 0; s < kRing; ++s)
      SERL_HIP(hipEventCreateWithFlags(&g_crop.done[s], hipEventDisableTiming));
    g_crop.device = device;
  }
  const size_t cbytes = sizeof(int32_t) * 2 * (size_t)batch;
  const size_t coff = (cbytes + 15) & ~(size_t)15;
  SERL_REQUIRE(2 * coff <= g_crop.slot_bytes, "batch too large");
  const int s = g_crop.next;
  g_crop.next = (s + 1) % kRing;
  if (g_crop.used[s]) SERL_HIP(hipEventSynchronize(g_crop.done[s]));
  uint8_t* h = g_crop.host + (size_t)s * g_crop.slot_bytes;
  uint8_t* d = g_crop.dev + (size_t)s * g_crop.slot_bytes;
  for (int k = 0; k < 2; ++k) {
    const int32_t* cr = k ? host_crop_next : host_crop_obs;
    if (cr) {
      for (int i = 0; i < 2 * batch; ++i)
        SERL_REQUIRE(cr[i] >= 0 && cr[i] <= 8, "crop offset %d out of [0,8]", cr[i]);
      std::memcpy(h + k * coff, cr, cbytes);
    }
  }
  SERL_HIP(hipMemcpyAsync(d, h, 2 * coff, hipMemcpyHostToDevice, stream));
  g_crop.used[s] = true;
  GatherArgs a{};
  for (int c = 0; c < n_cam; ++c) a.packed[c] = dev_packed[c];
  a.from_packed = 1;
  a.count0 = batch;
  a.batch = batch; a.n_cam = n_cam; a.H = H; a.W = W; a.C = C;
  a.crop_obs = host_crop_obs ? reinterpret_cast<const int32_t*>(d) : nullptr;
  a.crop_next = host_crop_next ? reinterpret_cast<const int32_t*>(d + coff) : nullptr;
  a.out_frames = dev_frames_out;
  int rc = launch_gather_crop(a, stream);
  if (rc) return rc;
  SERL_HIP(hipEventRecord(g_crop.done[s], stream));
  return SERL_OK;
}

}  // extern "C"
