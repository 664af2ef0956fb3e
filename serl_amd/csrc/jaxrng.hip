// JAX's PRNG in the product: threefry2x32 key chain on the host, normal / bernoulli draws on the device.
//
// The reference learner draws everything from `jax.random` under the default (threefry2x32, non-partitionable) implementation
// (SURVEY.md appendix B): the DrQ crop offsets (vision/data_augmentations.py:7-36: split(rng, B*T) then randint(key, (2,), 0, 9)
// per frame), the REDQ subsample (agents/continuous/sac.py:150-157: randint(key, (m,), 0, N)), the policy noise
// (sac.py:118-132,197-201,224-227: distrax sample = jax.random.normal(seed, (B, A))), the Dropout keep-masks
// (jax.random.bernoulli(key, 0.9, (B, 4096))) and the key bookkeeping of common/common.py:197-209 / sac.py:287-289 / drq.py:276-318.
// This file restates those functions -- jax/_src/prng.py (threefry_2x32, _threefry_split, _threefry_fold_in,
// _threefry_random_bits_original) and jax/_src/random.py (_uniform, _normal_real, _randint, _bernoulli), jax 0.4.x -- so that a
// learner started from the same seed consumes the numbers a JAX learner consumes:
//   * everything INTEGER (keys, crop offsets, REDQ indices, the 32-bit draws behind every sample) is bit-exact
//     (tests/test_jaxrng.py vs oracle/jaxshim/jax/threefry.py, which is pinned on the Random123 known-answer vectors);
//   * normal draws go bits -> uniform(-1, 1) -> sqrt(2) * erf_inv(u) with the single-precision polynomial XLA uses for f32
//     (Giles, "Approximating the erfinv function"; xla/client/lib/math.cc ErfInv32): equal to a JAX run up to the last bits of
//     log1p / sqrt.
// Host entry points are plain C (no device needed); the device entry points fill caller-provided HBM on a stream.
#include <cmath>
#include <cstdint>
#include <vector>

#include <hip/hip_runtime.h>

#include "common.h"
#include "jaxrng.h"
#include "../../include/serl_mi355.h"

namespace serl {

struct JaxJob { serl_jax_job j; };
constexpr int kMaxJaxJobs = 16;
struct JaxJobs { serl_jax_job j[kMaxJaxJobs]; long start[kMaxJaxJobs + 1]; int n; };

// one thread per output element: job found by its start offset; element e of the job's flat array
__global__ __launch_bounds__(256) void jax_fill_kernel(JaxJobs jobs) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= jobs.start[jobs.n]) return;
  int k = 0;
  while (k + 1 < jobs.n && t >= jobs.start[k + 1]) ++k;
  const serl_jax_job& j = jobs.j[k];
  const long i = t - jobs.start[k];
  const uint32_t b = random_bits_at(j.key[0], j.key[1], (uint64_t)j.n_total, (uint64_t)(j.first + i));
  if (j.kind == SERL_JAX_NORMAL) {
    static_cast<float*>(j.out)[i] = normal_from_bits(b);
  } else if (j.kind == SERL_JAX_BERNOULLI_U8) {
    static_cast<uint8_t*>(j.out)[i] = bits_to_unit(b) < j.p ? 1 : 0;
  } else {
    static_cast<uint32_t*>(j.out)[i] = b;
  }
}

static void split_host(const uint32_t key[2], int num, uint32_t* out /* [num][2] */) {
  const uint64_t n = 2ull * (uint64_t)num;
  for (uint64_t e = 0; e < n; ++e) out[e] = random_bits_at(key[0], key[1], n, e);
}

static void randint_host(const uint32_t key[2], int64_t n, int32_t lo, int32_t hi, int32_t* out) {
  uint32_t ks[4];
  split_host(key, 2, ks);
  const uint32_t span = (uint32_t)((int64_t)hi - (int64_t)lo > 0 ? (int64_t)hi - (int64_t)lo : 1);
  uint32_t mult = (uint32_t)(65536u % span);
  mult = (uint32_t)(mult * mult) % span;   // lax.mul on uint32 WRAPS (span > 65536: e.g. span 2^31 - 1 gives 0, not 2) -- jax._src.random._randint
  for (int64_t e = 0; e < n; ++e) {
    const uint32_t hb = random_bits_at(ks[0], ks[1], (uint64_t)n, (uint64_t)e);
    const uint32_t lb = random_bits_at(ks[2], ks[3], (uint64_t)n, (uint64_t)e);
    const uint32_t off = ((hb % span) * mult + (lb % span)) % span;   // uint32 arithmetic (wraps like lax.mul / lax.add)
    out[e] = (int32_t)((int64_t)lo + (int64_t)off);
  }
}

}  // namespace serl

extern "C" {

int serl_jax_prngkey(uint64_t seed, uint32_t key_out[2]) {
  SERL_REQUIRE(key_out, "NULL argument");
  // the reference runs JAX with x64 disabled: the seed is an int32, the high word of the key is always 0 (threefry_seed)
  key_out[0] = 0u;
  key_out[1] = (uint32_t)(seed & 0xFFFFFFFFull);
  return SERL_OK;
}

int serl_jax_split(const uint32_t key[2], int num, uint32_t* keys_out) {
  SERL_REQUIRE(key && keys_out && num >= 1, "bad argument");
  serl::split_host(key, num, keys_out);
  return SERL_OK;
}

int serl_jax_fold_in(const uint32_t key[2], uint32_t data, uint32_t key_out[2]) {
  SERL_REQUIRE(key && key_out, "NULL argument");
  // threefry_2x32(key, PRNGKey(data)) = the counter pair (0, data)
  uint32_t x0 = 0u, x1 = data;
  serl::threefry2x32(key[0], key[1], x0, x1);
  key_out[0] = x0; key_out[1] = x1;
  return SERL_OK;
}

int serl_jax_random_bits(const uint32_t key[2], int64_t n, uint32_t* out) {
  SERL_REQUIRE(key && out && n >= 0, "bad argument");
  for (int64_t e = 0; e < n; ++e) out[e] = serl::random_bits_at(key[0], key[1], (uint64_t)n, (uint64_t)e);
  return SERL_OK;
}

int serl_jax_randint(const uint32_t key[2], int64_t n, int32_t minval, int32_t maxval, int32_t* out) {
  SERL_REQUIRE(key && out && n >= 0, "bad argument");
  serl::randint_host(key, n, minval, maxval, out);
  return SERL_OK;
}

int serl_jax_normal_host(const uint32_t key[2], int64_t n, float* out) {
  SERL_REQUIRE(key && out && n >= 0, "bad argument");
  for (int64_t e = 0; e < n; ++e) out[e] = serl::normal_from_bits(serl::random_bits_at(key[0], key[1], (uint64_t)n, (uint64_t)e));
  return SERL_OK;
}

int serl_jax_crop_offsets(const uint32_t key[2], int frames, int padding, int32_t* yx_out) {
  SERL_REQUIRE(key && yx_out && frames >= 1 && padding >= 0, "bad argument");
  std::vector<uint32_t> ks(2 * (size_t)frames);
  serl::split_host(key, frames, ks.data());
  for (int i = 0; i < frames; ++i) serl::randint_host(&ks[2 * (size_t)i], 2, 0, 2 * padding + 1, yx_out + 2 * (size_t)i);
  return SERL_OK;
}

int serl_jax_update_keys(const uint32_t rng[2], int drq_aug, int n_critic, int has_actor_temp, int combined, serl_jax_update_keys_t* out) {
  SERL_REQUIRE(rng && out && n_critic >= 0 && n_critic <= SERL_JAX_MAX_UTD && (n_critic > 0 || has_actor_temp), "bad argument");
  SERL_REQUIRE(!combined || (n_critic == 1 && has_actor_temp), "a combined update is one critic + actor + temperature step");
  uint32_t r[2] = {rng[0], rng[1]};
  *out = serl_jax_update_keys_t{};
  if (drq_aug) {   // drq.py:276-277 / :307-308: rng, obs_rng, next_obs_rng = split(rng, 3); state.rng = rng
    uint32_t k[6];
    serl::split_host(r, 3, k);
    r[0] = k[0]; r[1] = k[1];
    out->k_obs[0] = k[2]; out->k_obs[1] = k[3];
    out->k_next[0] = k[4]; out->k_next[1] = k[5];
  }
  const int n_updates = combined ? 1 : n_critic + (has_actor_temp ? 1 : 0);
  for (int u = 0; u < n_updates; ++u) {
    // common.py:197-200: new_rng, *rngs = split(state.rng, 4) with the loss dict's leaves in sorted key order
    // (actor, critic, temperature); sac.py:287-289: afterwards state.rng = split(state.rng)[0] of the ENTRY rng
    uint32_t k[8];
    serl::split_host(r, 4, k);
    const uint32_t* r_actor = k + 2; const uint32_t* r_critic = k + 4; const uint32_t* r_temp = k + 6;
    if (u < n_critic) {
      uint32_t c[4];
      serl::split_host(r_critic, 2, c);           // sac.py:137: rng, next_action_sample_key = split(rng)
      out->k_next_action[u][0] = c[2]; out->k_next_action[u][1] = c[3];
      uint32_t s[4];
      serl::split_host(c, 2, s);                  // sac.py:151: rng, subsample_key = split(rng)
      out->k_subsample[u][0] = s[2]; out->k_subsample[u][1] = s[3];
    }
    if (combined || u >= n_critic) {
      uint32_t p[8];
      serl::split_host(r_actor, 4, p);            // sac.py:197: rng, policy_rng, sample_rng, critic_rng = split(rng, 4)
      out->k_policy[0] = p[2]; out->k_policy[1] = p[3];
      out->k_sample[0] = p[4]; out->k_sample[1] = p[5];
      uint32_t t[4];
      serl::split_host(r_temp, 2, t);             // sac.py:222: rng, next_action_sample_key = split(rng)
      out->k_temp[0] = t[2]; out->k_temp[1] = t[3];
    }
    uint32_t nx[4];
    serl::split_host(r, 2, nx);
    r[0] = nx[0]; r[1] = nx[1];
  }
  out->rng_out[0] = r[0]; out->rng_out[1] = r[1];
  out->n_critic = n_critic;
  return SERL_OK;
}

int serl_jax_fill(int device, const serl_jax_job* jobs, int n, void* stream) {
  SERL_REQUIRE(jobs && n >= 1 && n <= serl::kMaxJaxJobs, "1..16 jobs per launch");
  SERL_HIP(hipSetDevice(device));
  serl::JaxJobs jj{};
  long tot = 0;
  for (int i = 0; i < n; ++i) {
    const serl_jax_job& j = jobs[i];
    SERL_REQUIRE(j.out && j.n_total >= 0 && j.first >= 0 && j.count >= 0 && j.first + j.count <= j.n_total, "bad job %d", i);
    SERL_REQUIRE(j.kind == SERL_JAX_NORMAL || j.kind == SERL_JAX_BERNOULLI_U8 || j.kind == SERL_JAX_BITS, "bad job kind %d", j.kind);
    jj.j[i] = j;
    jj.start[i] = tot;
    tot += j.count;
  }
  jj.start[n] = tot;
  jj.n = n;
  if (tot == 0) return SERL_OK;
  hipLaunchKernelGGL(serl::jax_fill_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, jj);
  SERL_HIP(hipGetLastError());
  return SERL_OK;
}

}  // extern "C"
