/* Plain-C client of libserl_mi355.so: shows that the drop-in boundary needs nothing but the header (no torch,
 * no C++ types).  State-only SAC learner (BASELINE.json configs[0]): plain replay buffer in HBM -> fused gather ->
 * update_high_utd(utd_ratio = 2), three times.  Built by tests/test_abi.py's GPU test with
 *   gcc -std=c99 tests/c_abi_smoke.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lserl_amd/lib -lserl_mi355 -L/opt/rocm/lib -lamdhip64 -lm
 * Prints "C ABI OK" and exits 0 on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "serl_mi355.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, serl_last_error()); return 1; } \
  } while (0)
#define HIPCHECK(call)                                                               \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)

static unsigned lcg_state = 12345u;
static float frand(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) / 16777216.0f - 0.5f; }

int main(void) {
  enum { S = 10, A = 4, B = 64, CAP = 500 };
  if (serl_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 2; }
  serl_rb* rb = NULL;
  CHECK(serl_rb_create(0, CAP, 0, 0, 0, 0, 1, S, A, &rb));
  CHECK(serl_rb_seed(rb, 0x0123456789abcdefULL, 0xfedcba9876543210ULL, 0x1ULL, 0x2545F4914F6CDD1DULL | 1ULL, 0, 0));
  float st[S], nst[S], act[A];
  for (int t = 0; t < 300; ++t) {
    for (int i = 0; i < S; ++i) { st[i] = frand(); nst[i] = frand(); }
    for (int i = 0; i < A; ++i) act[i] = 2.0f * frand();
    CHECK(serl_rb_insert(rb, NULL, NULL, st, nst, act, (t % 20 == 19) ? 1.0f : 0.0f, (t % 20 == 19) ? 0.0f : 1.0f, t % 20 == 19));
  }
  if (serl_rb_len(rb) != 300) { fprintf(stderr, "len %lld\n", (long long)serl_rb_len(rb)); return 1; }

  serl_agent_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.device = 0; cfg.n_cam = 0; cfg.state_dim = S; cfg.act_dim = A; cfg.batch = B; cfg.ensemble = 10;
  cfg.hidden = 256; cfg.bottleneck = 256; cfg.sle_features = 8; cfg.proprio_dim = 64;
  cfg.warmup_steps = 4; cfg.temp_warmup_steps = 0; cfg.discount = 0.99f; cfg.tau = 0.005f; cfg.lr = 3e-4f;
  cfg.dropout = 0.1f; cfg.std_min = 1e-5f; cfg.std_max = 5.0f; cfg.target_entropy = -A / 2.0f; cfg.seed = 7;
  serl_agent* ag = NULL;
  CHECK(serl_agent_create(&cfg, &ag));
  /* parameters: small random kernels, unit LayerNorm scales (leaf names from the library) */
  char name[128];
  int64_t cnt = 0;
  for (int i = 0; i < serl_agent_num_leaves(ag); ++i) {
    CHECK(serl_agent_leaf_info(ag, i, name, sizeof name, &cnt));
    if (strncmp(name, "trunk/", 6) == 0) continue;
    float* v = (float*)malloc(sizeof(float) * (size_t)cnt);
    const int is_scale = strstr(name, "/scale") != NULL, is_kernel = strstr(name, "/w") != NULL || strstr(name, "kernel") != NULL;
    for (int64_t k = 0; k < cnt; ++k) v[k] = is_scale ? 1.0f : (is_kernel ? 0.1f * frand() : 0.0f);
    if (strcmp(name, "temp/lagrange") == 0) v[0] = logf(expf(0.01f) - 1.0f);
    CHECK(serl_agent_set(ag, "params", name, v, cnt));
    CHECK(serl_agent_set(ag, "target_params", name, v, cnt));
    free(v);
  }

  serl_batch db;
  memset(&db, 0, sizeof db);
  db.batch = B; db.n_cam = 0; db.state_dim = S; db.act_dim = A;
  HIPCHECK(hipMalloc((void**)&db.state, sizeof(float) * 2 * B * S));
  HIPCHECK(hipMalloc((void**)&db.action, sizeof(float) * B * A));
  HIPCHECK(hipMalloc((void**)&db.reward, sizeof(float) * B));
  HIPCHECK(hipMalloc((void**)&db.mask, sizeof(float) * B));
  HIPCHECK(hipMalloc((void**)&db.done, B));
  hipStream_t stream;
  HIPCHECK(hipStreamCreate(&stream));
  int64_t idx[B];
  serl_info info;
  for (int it = 0; it < 3; ++it) {
    CHECK(serl_rb_sample_indices(rb, B, idx));
    int64_t* idxp = idx;   /* in/out: stale indices are re-drawn in place */
    const int count = B;
    serl_rb* rbs[1] = {rb};
    CHECK(serl_rb_gather_crop(rbs, 1, &idxp, &count, NULL, NULL, &db, stream));
    CHECK(serl_agent_update_high_utd(ag, &db, 2, NULL, stream));
    CHECK(serl_agent_read_info(ag, &info, stream));
    if (!isfinite(info.critic_loss) || !isfinite(info.actor_loss) || !isfinite(info.temperature_loss)) {
      fprintf(stderr, "non-finite info\n");
      return 1;
    }
  }
  if (serl_agent_get_step(ag) != 9) { fprintf(stderr, "step %lld\n", (long long)serl_agent_get_step(ag)); return 1; }
  printf("critic_loss %.5f actor_loss %.5f temperature %.5f\n", info.critic_loss, info.actor_loss, info.temperature);
  CHECK(serl_agent_destroy(ag));
  CHECK(serl_rb_destroy(rb));
  printf("C ABI OK\n");
  return 0;
}
