"""Copies gpurun_out/evidence (scripts/collect_evidence.sh) into profiles/ as r04_* and rewrites the 'Round 4, final state'
table of profiles/README.md from the JSON files (history / rejected-experiment sections are kept as they are).
Extra evidence directories given on the command line (scripts/collect_evidence_min.sh, collect_evidence_rest.sh) are laid over
it; a file that no directory holds keeps its tracked profiles/ copy and is listed as KEPT (printed, and named in the table's
last row) so that the table never silently mixes calls."""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, "gpurun_out", "evidence")
P = os.path.join(R, "profiles")
RD = "r04"
pairs = {"bench": "bench", "bench_f32": "bench_trunk_f32", "bench_car4": "bench_car4", "bench_serial": "bench_serial",
         "bench_unfused_gn": "bench_unfused_gn", "bench_gemm_f32": "bench_gemm_f32", "bench_emulate_world2": "bench_emulate_world2",
         "bench_emulate_world4": "bench_emulate_world4", "bench_emulate_world8": "bench_emulate_world8",
         "bench_drq_demos": "bench_drq_demos", "bench_peg": "bench_peg", "bench_fwbw": "bench_fwbw",
         "bench_small_encoder": "bench_small_encoder", "bench_collective_1rank": "bench_collective_1rank",
         "bench_emulate_world8_collective": "bench_emulate_world8_collective",
         "bench_chain_unfused": "bench_chain_unfused", "bench_chain_unfusedemulateworld8": "bench_chain_unfused_emulate_world8",
         "bench_chain_unfusednopipeline": "bench_chain_unfused_serial",
         "bench_chain_lnepi": "bench_chain_lnepi", "bench_chain_lnepiemulateworld8": "bench_chain_lnepi_emulate_world8",
         "bench_chain_lnepinopipeline": "bench_chain_lnepi_serial",
         "actor_latency": "actor_latency", "sac_state": "sac_state"}
DIRS = [os.path.join(R, "gpurun_out", d) for d in sys.argv[1:]] + [E]
KEPT = []
def find(name):
    for d in DIRS:
        q = os.path.join(d, name)
        if os.path.exists(q) and os.path.getsize(q) > 0:
            return q
    return None
J = {}
for src, dst in pairs.items():
    q = find(src + ".json")
    if q is None:
        KEPT.append(f"{RD}_{dst}.json")
        J[dst] = json.load(open(os.path.join(P, f"{RD}_{dst}.json")))
        continue
    d = json.loads(open(q).read().strip().splitlines()[-1])
    json.dump(d, open(os.path.join(P, f"{RD}_{dst}.json"), "w"), indent=1)
    J[dst] = d
for src, dst in (("kernel_stats.csv", f"{RD}_kernel_stats.csv"), ("kernel_stats_serial.csv", f"{RD}_kernel_stats_serial.csv"),
                 ("kernel_stats_small.csv", f"{RD}_kernel_stats_small_encoder.csv"),
                 ("pmc_traffic.json", "pmc_traffic.json"), ("mfma_counters.json", f"{RD}_mfma_counters.json"),
                 ("wait_counters.json", f"{RD}_wait_counters.json"), ("frac_from_stats.txt", f"{RD}_frac_from_stats.txt"),
                 ("timeline.txt", f"{RD}_timeline.txt"), ("launches_pipelined.txt", f"{RD}_launches_pipelined.txt"),
                 ("launches_serial.txt", f"{RD}_launches_serial.txt"), ("launches_emulate_world8.txt", f"{RD}_launches_emulate_world8.txt")):
    q = find(src)
    if q is None:
        KEPT.append(dst)
        continue
    shutil.copy(q, os.path.join(P, dst))
b, f32, c4, se, un, gf = (J[k] for k in ("bench", "bench_trunk_f32", "bench_car4", "bench_serial", "bench_unfused_gn", "bench_gemm_f32"))
e2, e4, e8 = J["bench_emulate_world2"], J["bench_emulate_world4"], J["bench_emulate_world8"]
w2, w3, w4, sm, co = J["bench_drq_demos"], J["bench_peg"], J["bench_fwbw"], J["bench_small_encoder"], J["bench_collective_1rank"]
al, sac = J["actor_latency"], J["sac_state"]
pm = json.load(open(os.path.join(P, "pmc_traffic.json")))
mc = json.load(open(os.path.join(P, f"{RD}_mfma_counters.json")))
wc = json.load(open(os.path.join(P, f"{RD}_wait_counters.json")))
fr = open(os.path.join(P, f"{RD}_frac_from_stats.txt")).read().strip().splitlines()
r, cb = b["roofline"], b["cpu_baseline"]
g = lambda d, k: d.get(k, {})
runs = lambda d: " / ".join(str(x) for x in d.get("ms_per_step_runs", []))
ver = lambda d: (d.get("verify") or {}).get("worst_rel_diff")
coll = co.get("collective", {})
rows = f"""| file | what | command |
|---|---|---|
| `{RD}_bench.json` | official bench line (replay cap 200k / fill 20k, CAR=1, pipelined, split-fp16 trunk, median of 3 x 200 timed steps: {runs(b)} ms): **{b['value']} grad-steps/s** ({b['ms_per_step']} ms/step). Block-conv family (11 launches per trunk pass, HIP events around every 16th launch inside the timed region (`roofline.note`), co-running with the update chain; stage-0/1 durations include their fused GroupNorm epilogues): {r['algorithmic_tflops']} algorithmic TFLOP/s = {r['achieved']} TFLOP/s of executed fp16 MFMA = **{100*r['frac']:.1f} %** of the 2.5 PF dense peak (per stage: {r.get('frac_by_stage')}); post-run verification of the co-running fused epilogues: worst relative difference {ver(b):.2e} over {b['verify']['batches_checked']} batches (tolerance {b['verify']['tol']}); `gather_crop_rgb` {r['sample_aug_hbm']['achieved']} TB/s co-running ({se['roofline']['sample_aug_hbm']['achieved']} TB/s alone, `{RD}_bench_serial.json`); CPU port {cb['value']} grad-steps/s on {cb['cores']} cores -> {b['value']/cb['value']:.0f}x | `python bench.py` |
| `{RD}_bench_serial.json` | no overlap of trunk(i+1) with update(i): {se['value']} grad-steps/s ({se['ms_per_step']} ms = trunk + update chain); per-kernel times here are uncontended: block convs {se['roofline']['achieved']} TFLOP/s executed = {100*se['roofline']['frac']:.1f} % | `python bench.py --no-pipeline --no-cpu-baseline --steps 110` |
| `{RD}_bench_unfused_gn.json` | official line with the GroupNorm epilogues switched off (separate elementwise passes): {un['value']} grad-steps/s ({un['ms_per_step']} ms); the conv kernels alone then run at {100*un['roofline']['frac']:.1f} % | `SERL_GN_FUSE=0 python bench.py --no-cpu-baseline --steps 110` |
| `{RD}_bench_gemm_f32.json` | official line with the update chain's GEMMs on the exact fp32 MFMA kernel instead of the bf16x3 one: {gf['value']} grad-steps/s ({gf['ms_per_step']} ms) | `SERL_GEMM=f32 python bench.py --no-cpu-baseline --steps 110` |
| `{RD}_bench_trunk_f32.json` | exact-fp32 MFMA trunk: {f32['value']} grad-steps/s, conv family {f32['roofline']['achieved']} TFLOP/s = {100*f32['roofline']['frac']:.1f} % of the 157.3 TFLOP/s fp32-MFMA peak | `python bench.py --trunk f32 --no-cpu-baseline --steps 40` |
| `{RD}_bench_car4.json` | critic_actor_ratio 4 (the reference script's default): {c4['value']} grad-steps/s | `python bench.py --car 4 --no-cpu-baseline --steps 50` |
| `{RD}_bench_drq_demos.json`, `{RD}_bench_peg.json`, `{RD}_bench_fwbw.json` | BASELINE.json configs[2..4] as bench workloads (two HBM replay buffers sampled 50/50 and concatenated on the device; CAR 8 / 8 / 4; batch 256 / 256 / 512; a step = one `update_high_utd` call = CAR grad steps): **{w2['value']} / {w3['value']} / {w4['value']} grad-steps/s** ({w2['ms_per_step']} / {w3['ms_per_step']} / {w4['ms_per_step']} ms per call); verification {ver(w2):.1e} / {ver(w3):.1e} / {ver(w4):.1e} | `python bench.py --workload drq_demos` (`peg`, `fwbw`) |
| `{RD}_bench_small_encoder.json` | `encoder_type="small"` (trainable SmallEncoder, forward + backward through the encoder every grad step, no frozen trunk): {sm['value']} grad-steps/s ({sm['ms_per_step']} ms); conv stack (implicit GEMMs on the bf16x3 kernel) at {sm['roofline'].get('algorithmic_tflops', sm['roofline']['achieved'])} algorithmic TFLOP/s; round 2 with explicit im2col matrices: 68.0 grad-steps/s | `python bench.py --encoder small --no-cpu-baseline --steps 40` |
| `{RD}_bench_emulate_world{{2,4,8}}.json` | ONE rank's share (B/N samples, no collective) of an N-GPU data-parallel step on this GPU = upper bound of the strong-scaling step rate before RCCL time: {e2['value']} / {e4['value']} / {e8['value']} grad-steps/s ({e2['ms_per_step']} / {e4['ms_per_step']} / {e8['ms_per_step']} ms) -> {e2['value']/b['value']:.2f}x / {e4['value']/b['value']:.2f}x / {e8['value']/b['value']:.2f}x of 1 GPU | `python bench.py --emulate-world N --steps 110 --no-cpu-baseline` |
| `{RD}_bench_chain_unfused*.json`, `{RD}_bench_chain_lnepi*.json` | the update chain's variants in the SAME call as the official line (pipelined / one rank's share of 8 / serial): default (fused launches, separate LayerNorm launches: 48 per critic + actor pair) {b['ms_per_step']} / {e8['ms_per_step']} / {se['ms_per_step']} ms; one launch per operation (`SERL_CHAIN_FUSE=0`, the round-3 schedule, 63 launches) {J['bench_chain_unfused']['ms_per_step']} / {J['bench_chain_unfused_emulate_world8']['ms_per_step']} / {J['bench_chain_unfused_serial']['ms_per_step']} ms; LayerNorm + tanh inside the GEMM launches as well (`SERL_CHAIN_LN_EPI=1`, 38 launches) {J['bench_chain_lnepi']['ms_per_step']} / {J['bench_chain_lnepi_emulate_world8']['ms_per_step']} / {J['bench_chain_lnepi_serial']['ms_per_step']} ms; with the RCCL calls really issued on a 1-rank group at B/8: {J['bench_emulate_world8_collective']['ms_per_step']} ms | `SERL_CHAIN_FUSE=0 / SERL_CHAIN_LN_EPI=1 python bench.py [--emulate-world 8] [--no-pipeline] --steps 110 --no-cpu-baseline` |
| `{RD}_bench_collective_1rank.json` | the N > 1 code path on one rank (RCCL all-reduces really issued, world size 1): {co['value']} grad-steps/s; {coll.get('all_reduces_per_step')} all-reduces per step, {coll.get('bytes_per_step')} bytes; per all-reduce {coll.get('avg_us_by_bytes')} us | `python bench.py --force-collective --no-cpu-baseline --steps 110` |
| `{RD}_kernel_stats.csv`, `{RD}_kernel_stats_serial.csv`, `{RD}_kernel_stats_small_encoder.csv` | `rocprofv3 --kernel-trace --stats` per-kernel summaries of the pipelined, the serial and the SmallEncoder bench commands (`scripts/rocprof_summary.py`) | `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-verify [--no-pipeline] [--encoder small] --fill 3000 --steps 30 --warmup 5 --repeats 1` |
| `{RD}_frac_from_stats.txt` | `roofline.frac` recomputed from those CSVs alone (`scripts/frac_from_stats.py`): pipelined `{fr[0].split('frac')[-1].strip()}` vs {r['frac']} from the HIP events inside `bench.py`; serial `{fr[1].split('frac')[-1].strip()}` vs {se['roofline']['frac']} | `python scripts/frac_from_stats.py profiles/{RD}_kernel_stats_serial.csv` |
| `{RD}_mfma_counters.json` | counter-based MFMA utilisation and LDS bank conflicts per kernel family (two `--pmc` passes, serial schedule; `scripts/pmc_counters.py`): matrix pipe busy per SIMD: conv_dma {g(mc,'conv_dma_f16x3').get('mfma_util_per_simd')}, row-slab {g(mc,'conv3x3_rowslab_f16x3').get('mfma_util_per_simd')}, conv_init_u8 {g(mc,'conv_init_u8').get('mfma_util_per_simd')}, gemm_bf16x3 {g(mc,'gemm_bf16x3').get('mfma_util_per_simd')}; LDS cycles lost to bank conflicts: conv_dma {g(mc,'conv_dma_f16x3').get('lds_conflict_frac')}, row-slab {g(mc,'conv3x3_rowslab_f16x3').get('lds_conflict_frac')}, conv_init_u8 {g(mc,'conv_init_u8').get('lds_conflict_frac')}, gemm_bf16x3 {g(mc,'gemm_bf16x3').get('lds_conflict_frac')} | `scripts/collect_evidence.sh` |
| `{RD}_wait_counters.json` | where the waves' time goes (`--pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES`, `scripts/pmc_wait.py`): share of a wave's lifetime parked on s_waitcnt / barriers: conv_dma {g(wc,'conv_dma_f16x3').get('wait_any_frac')}, row-slab {g(wc,'conv3x3_rowslab_f16x3').get('wait_any_frac')}, conv_init_u8 {g(wc,'conv_init_u8').get('wait_any_frac')}, gemm_bf16x3 {g(wc,'gemm_bf16x3').get('wait_any_frac')}, pool_finish {g(wc,'pool_finish_split').get('wait_any_frac')} | `scripts/collect_evidence.sh` |
| `pmc_traffic.json` | HBM traffic per launch from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, serial schedule, `scripts/pmc_to_json.py`; FETCH doubled per the gfx950 note). Block-conv family: {pm['conv_igemm_f16x3_bytes_per_launch']/1e6:.0f} MB per launch (fetch {pm['conv_igemm_f16x3_fetch_bytes_per_launch']/1e6:.0f} MB, write {pm['conv_igemm_f16x3_write_bytes_per_launch']/1e6:.0f} MB); gather_crop_rgb {pm['gather_crop_bytes_per_launch']/1e6:.1f} MB measured vs 100.72 MB algorithmic; conv_init_u8 (which now completes the pooling: no pool-finish pass) {pm['conv_init_f16x3_bytes_per_launch']/1e6:.0f} MB per pass (fetch {pm['conv_init_f16x3_fetch_bytes_per_launch']/1e6:.0f}, write {pm['conv_init_f16x3_write_bytes_per_launch']/1e6:.0f}) against 318 MB algorithmic (50 MB of u8 frames in, 268 MB of pooled fp32 out) = 1.6x; round 2: conv_init 457 MB + pool finish 640 MB = 3.4x | `scripts/collect_evidence.sh` |
| `{RD}_timeline.txt`, `{RD}_launches_pipelined.txt`, `{RD}_launches_serial.txt`, `{RD}_launches_emulate_world8.txt` | one step from a kernel trace: both streams, start offsets and durations (`scripts/timeline_full.py`); per launch: kernel, workgroups, VGPRs, duration, idle time of its queue before it (`scripts/chain_trace.py`) for the pipelined, the serial and the B/8 schedule | `rocprofv3 --kernel-trace -- python bench.py ... --steps 12 [--no-pipeline] [--emulate-world 8]` |
| `{RD}_actor_latency.json` | next-row N3: `agent.sample_actions` on ONE observation: {al['host_ms_per_call']} ms per call on the host, {al['device_ms_per_call']} ms of device time -> {al['actions_per_s']} actions/s | `python bench.py --workload actor_latency` |
| `{RD}_sac_state.json` | BASELINE.json configs[0] `async_sac_state_sim` (state-only SAC, 2048 = 256 x UTD 8 per iteration): {sac['critic_grad_steps_per_s']} critic grad-steps/s ({sac['ms_per_iteration']} ms per iteration) vs {sac['cpu_port']['critic_grad_steps_per_s']} on {sac['cpu_port']['cores']} CPU cores (oracle port) -> {sac['speedup']}x | `python bench.py --workload sac_state --steps 200` |
"""
sm_pmc = find("mfma_counters_small_encoder.json")
if sm_pmc:
    shutil.copy(sm_pmc, os.path.join(P, f"{RD}_mfma_counters_small_encoder.json"))
    sc = json.load(open(sm_pmc))
    rows += (f"| `{RD}_mfma_counters_small_encoder.json` | the same two counter passes on the SmallEncoder bench command (`--encoder small`, serial): "
             + "; ".join(f"{k}: matrix pipe busy {v.get('mfma_util_per_simd')}, LDS conflict share {v.get('lds_conflict_frac')}" for k, v in sc.items() if isinstance(v, dict))
             + " | `scripts/collect_evidence_rest.sh` |\n")
if os.path.exists(os.path.join(P, f"{RD}_pytest_gpu.txt")):
    head = open(os.path.join(P, f"{RD}_pytest_gpu.txt")).readline().lstrip("# ").strip()
    rows += f"| `{RD}_pytest_gpu.txt` | {head} | `scripts/collect_evidence_min.sh`, `scripts/collect_evidence_tests.sh` |\n"
if KEPT:
    rows += ("| (kept) | NOT re-measured by the last evidence call(s) -- these files are from the earlier round-4 call on commit 1dc8a4b "
             "(before the conv_init load / store reordering, the SmallEncoder loaders and the Adam kernel changed): "
             + ", ".join(f"`{k}`" for k in KEPT) + " | |\n")
p = os.path.join(P, "README.md")
s = open(p).read()
a = s.index(f"<!-- {RD}-table-begin -->") + len(f"<!-- {RD}-table-begin -->\n")
z = s.index(f"<!-- {RD}-table-end -->")
s = s[:a] + rows + s[z:]
open(p, "w").write(s)
print("profiles refreshed:", b["value"], "grad-steps/s; frac", r["frac"])
