"""Copies gpurun_out/evidence (scripts/collect_evidence.sh) into profiles/ and rewrites the 'final state' table of
profiles/README.md from the JSON files (the history / rejected-experiments sections are kept as they are)."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, "gpurun_out", "evidence")
P = os.path.join(R, "profiles")
pairs = {"bench_f16x3": "r01_bench", "bench_f32": "r01_bench_trunk_f32", "bench_f16x3_car4": "r01_bench_car4",
         "bench_serial": "r01_bench_serial", "bench_emulate_world2": "r01_bench_emulate_world2",
         "bench_emulate_world4": "r01_bench_emulate_world4", "bench_emulate_world8": "r01_bench_emulate_world8"}
J = {}
for src, dst in pairs.items():
    d = json.loads(open(os.path.join(E, src + ".json")).read().strip().splitlines()[-1])
    json.dump(d, open(os.path.join(P, dst + ".json"), "w"), indent=1)
    J[dst] = d
shutil.copy(os.path.join(E, "kernel_stats.csv"), os.path.join(P, "r01_kernel_stats.csv"))
shutil.copy(os.path.join(E, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
b, f32, c4, se = J["r01_bench"], J["r01_bench_trunk_f32"], J["r01_bench_car4"], J["r01_bench_serial"]
e2, e4, e8 = J["r01_bench_emulate_world2"], J["r01_bench_emulate_world4"], J["r01_bench_emulate_world8"]
pm = json.load(open(os.path.join(P, "pmc_traffic.json")))
r, cb = b["roofline"], b["cpu_baseline"]
sac = json.load(open(os.path.join(P, "r01_sac_state.json"))) if os.path.exists(os.path.join(P, "r01_sac_state.json")) else None
rows = f"""| file | what | command |
|---|---|---|
| `r01_bench.json` | official bench line (replay cap 200k / fill 20k, CAR=1, pipelined, split-fp16 trunk): **{b['value']} grad-steps/s** ({b['ms_per_step']} ms/step). conv family (11 launches per trunk pass, HIP events around every 4th launch inside the timed region, co-running with the update chain): {r['algorithmic_tflops']} algorithmic TFLOP/s = {r['achieved']} TFLOP/s of executed fp16 MFMA = **{100*r['frac']:.1f} %** of the 2.5 PF dense peak ({r['algorithmic_vs_f32_mfma_peak']}x the fp32-MFMA peak); `gather_crop` {r['sample_aug_hbm']['achieved']} TB/s co-running ({se['roofline']['sample_aug_hbm']['achieved']} TB/s alone, `r01_bench_serial.json`); CPU port {cb['value']} grad-steps/s on {cb['cores']} cores ({cb['cpu']}) -> {b['value']/cb['value']:.0f}x | `python bench.py` |
| `r01_bench_trunk_f32.json` | same with the exact-fp32 MFMA trunk: {f32['value']} grad-steps/s, conv family {f32['roofline']['achieved']} TFLOP/s = {100*f32['roofline']['frac']:.1f} % of the 157.3 TFLOP/s fp32-MFMA peak | `python bench.py --trunk f32 --no-cpu-baseline` |
| `r01_bench_car4.json` | critic_actor_ratio 4 (the reference script's default): {c4['value']} grad-steps/s | `python bench.py --car 4 --no-cpu-baseline` |
| `r01_bench_serial.json` | no overlap of trunk(i+1) with update(i): {se['value']} grad-steps/s ({se['ms_per_step']} ms = trunk + update chain); per-kernel times here are uncontended: conv family {se['roofline']['achieved']} TFLOP/s executed = {100*se['roofline']['frac']:.1f} % | `python bench.py --no-pipeline --no-cpu-baseline` |
| `r01_bench_emulate_world{{2,4,8}}.json` | ONE rank's share (B/N samples, no collective) of an N-GPU data-parallel step on this GPU = upper bound of the strong-scaling step rate before RCCL time: {e2['value']} / {e4['value']} / {e8['value']} grad-steps/s ({e2['ms_per_step']} / {e4['ms_per_step']} / {e8['ms_per_step']} ms) -> {e2['value']/b['value']:.2f}x / {e4['value']/b['value']:.2f}x / {e8['value']/b['value']:.2f}x of 1 GPU | `python bench.py --emulate-world N --steps 100 --no-cpu-baseline` |
"""
if sac:
    rows += f"| `r01_sac_state.json` | side measurement of BASELINE.json configs[0] `async_sac_state_sim` (state-only SAC, 2048 = 256 x UTD 8 per iteration, plain replay buffer in HBM): {sac['critic_grad_steps_per_s']} critic grad-steps/s ({sac['ms_per_iteration']} ms per iteration of 8 critic + 1 actor/temperature updates) vs {sac['cpu_port']['critic_grad_steps_per_s']} on {sac['cpu_port']['cores']} CPU cores (oracle port) -> {sac['speedup']}x; latency-bound (small MLPs: ~8 us per dependent kernel) | `python bench.py --workload sac_state --steps 200` |\n"
rows += f"""| `r01_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats` per-kernel summary of the bench workload (pipelined, so the update chain's small kernels co-run with trunk kernels and are stretched; `__amd_rocclr_copyBuffer` = replay inserts of the fill phase); agrees with the live HIP events of `r01_bench.json` within the profiler's overhead | `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --fill 3000 --steps 30 --warmup 5` + `scripts/rocprof_summary.py` |
| `pmc_traffic.json` | HBM traffic per launch from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, serial schedule, `scripts/pmc_to_json.py`; FETCH doubled per the gfx950 note). conv family: {pm['conv_igemm_f16x3_bytes_per_launch']/1e6:.0f} MB per launch (fetch {pm['conv_igemm_f16x3_fetch_bytes_per_launch']/1e6:.0f} MB, write {pm['conv_igemm_f16x3_write_bytes_per_launch']/1e6:.0f} MB); gather_crop {pm['gather_crop_bytes_per_launch']/1e6:.1f} MB measured vs 100.72 MB algorithmic (no wasted re-reads); conv_init {pm['conv_init_f16x3_bytes_per_launch']/1e6:.0f} MB and pool stage {pm['gn_relu_maxpool_bytes_per_launch']/1e6:.0f} MB per pass with the fused pool (1199 MB and 1888 MB before it) | `scripts/collect_evidence.sh` |
| `r01_timeline_emulate_world8.txt` | per-stream kernel timeline of one step at B/8 (`scripts/timeline_streams.py` on a kernel trace): trunk stream 0.62 ms busy in 24 kernels, update stream 0.56 ms busy in 62 kernels (the trace itself makes the host the bottleneck: 1.02 ms per step under rocprofv3 vs 0.75 ms without) | |
"""
p = os.path.join(P, "README.md")
s = open(p).read()
a = s.index("| file | what | command |")
z = s.index("## Round 1, history")
s = s[:a] + rows + "\n" + s[z:]
open(p, "w").write(s)
print("profiles refreshed:", b["value"], "grad-steps/s; frac", r["frac"])
