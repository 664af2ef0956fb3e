"""Copies gpurun_out/r06_evidence (scripts/r06_evidence.sh) into profiles/ as r06_* (plus pmc_traffic.json and scaling_pieces.json, which bench.py
reads) and rewrites the 'Round 6, final state' table of profiles/README.md between the r06-table markers; the commit the evidence ran on
(commit.txt) goes into the table.  Files the evidence call did not produce
are reported, never silently replaced."""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(R, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r06_evidence")
P = os.path.join(R, "profiles")
RD = "r06"
J, missing = {}, []
for f in sorted(os.listdir(E)):
    if f.endswith(".json") and (f.startswith("bench") or f in ("actor_latency.json", "sac_state.json")):
        try:
            d = json.loads(open(os.path.join(E, f)).read().strip().splitlines()[-1])
        except Exception:
            missing.append(f)
            continue
        name = f[:-5]
        json.dump(d, open(os.path.join(P, f"{RD}_{name}.json"), "w"), indent=1)
        J[name] = d
for src, dst in (("kernel_stats.csv", f"{RD}_kernel_stats.csv"), ("kernel_stats_serial.csv", f"{RD}_kernel_stats_serial.csv"),
                 ("kernel_stats_small.csv", f"{RD}_kernel_stats_small_encoder.csv"), ("pmc_traffic.json", "pmc_traffic.json"),
                 ("mfma_counters.json", f"{RD}_mfma_counters.json"), ("wait_counters.json", f"{RD}_wait_counters.json"),
                 ("frac_from_stats.txt", f"{RD}_frac_from_stats.txt"), ("timeline.txt", f"{RD}_timeline.txt"),
                 ("timeline_streams.txt", f"{RD}_timeline_streams.txt"), ("launches_pipelined.txt", f"{RD}_launches_pipelined.txt"),
                 ("launches_serial.txt", f"{RD}_launches_serial.txt"), ("launches_farm_updater.txt", f"{RD}_launches_farm_updater.txt"),
                 ("scaling_pieces.json", "scaling_pieces.json"), ("commit.txt", f"{RD}_evidence_commit.txt")):
    q = os.path.join(E, src)
    if os.path.exists(q) and os.path.getsize(q) > 0:
        shutil.copy(q, os.path.join(P, dst))
    else:
        missing.append(src)
print("missing:", missing)

def ms(n):
    d = J.get(n)
    return "n/a" if d is None else str(d.get("ms_per_step", d.get("diagnostic_ms_per_step")))
def val(n):
    d = J.get(n)
    return "n/a" if d is None else str(d.get("value", "-"))
b = J["bench"]; r = b["roofline"]; cb = b.get("cpu_baseline", {})
pk = r["per_kernel"]
us = lambda t: pk.get(t, {}).get("avg_us", float("nan"))
tr = json.load(open(os.path.join(P, "pmc_traffic.json"))) if os.path.exists(os.path.join(P, "pmc_traffic.json")) else {}
commit = open(os.path.join(E, "commit.txt")).read().strip() if os.path.exists(os.path.join(E, "commit.txt")) else "unknown"
sp = json.load(open(os.path.join(P, "scaling_pieces.json"))) if os.path.exists(os.path.join(P, "scaling_pieces.json")) else {}
rows = [
 f"| `{RD}_evidence_commit.txt` | every file of this table was measured on commit **{commit}** in ONE gpurun call (`scripts/r06_evidence.sh`) |",
 f"| `{RD}_bench.json` | official bench line (`python bench.py`: replay cap 200k / fill 20k, CAR 1, pipelined, split-fp16 trunk, jax.random stream drawn inside the kernels; median of 3 x {b['steps']} timed steps: {b['ms_per_step_runs']} ms): **{b['value']} grad-steps/s** ({b['ms_per_step']} ms/step).  Block-conv family ({r['kernel'].split(':')[0].split(', ')[-1]}): {r['algorithmic_tflops']} algorithmic TFLOP/s = {r['achieved']} TFLOP/s executed fp16 MFMA = **{100*r['frac']:.1f} %** of the 2.5 PF dense peak, per stage {r.get('frac_by_stage')}; per pass: conv_init {us('conv_init'):.0f}, b0 {us('conv_igemm/b0_conv0'):.0f} + {us('conv_igemm/b0_conv1'):.0f}, b1 {us('conv_igemm/b1_conv0'):.0f} (with its projection) + {us('conv_igemm/b1_conv1'):.0f}, b2 {us('conv_igemm/b2_conv0'):.0f} + {us('conv_igemm/b2_conv1'):.0f}, b3 {us('conv_igemm/b3_conv0'):.0f} + {us('conv_igemm/b3_conv1'):.0f} us; whole step {r.get('whole_step')}; verification {b.get('verify')}; gather_crop {r.get('sample_aug_hbm')}; CPU port {cb.get('value')} grad-steps/s on {cb.get('cores')} cores -> {b['value']/cb['value'] if cb.get('value') else float('nan'):.0f}x |",
 f"| `{RD}_bench_serial.json` | no overlap of trunk(i+1) with update(i): {val('bench_serial')} grad-steps/s ({ms('bench_serial')} ms); block convs uncontended: frac {J.get('bench_serial', {}).get('roofline', {}).get('frac')} |",
 f"| `{RD}_bench_emulate_world{{2,4,8}}.json` | one rank's share of an N-GPU batch-sharded step (no collective): {ms('bench_emulate_world2')} / {ms('bench_emulate_world4')} / {ms('bench_emulate_world8')} ms; with the RCCL calls on a 1-rank group at B/8: {ms('bench_emulate_world8_collective')} ms |",
 f"| `{RD}_bench_farm_worker.json`, `{RD}_bench_farm_updater.json` | the trunk farm's two roles alone on this GPU (diagnostic lines, no `metric`): worker pass {ms('bench_farm_worker')} ms, updater step {ms('bench_farm_updater')} ms |",
 f"| `{RD}_bench_drq_demos.json`, `{RD}_bench_peg.json`, `{RD}_bench_fwbw.json` | BASELINE configs[2..4]: {val('bench_drq_demos')} / {val('bench_peg')} / {val('bench_fwbw')} grad-steps/s |",
 f"| `{RD}_bench_small_encoder.json` | `--encoder small`: {val('bench_small_encoder')} grad-steps/s ({ms('bench_small_encoder')} ms) |",
 f"| `{RD}_bench_car4.json`, `{RD}_bench_collective_1rank.json` | CAR 4: {val('bench_car4')}; the N > 1 code path on one rank (RCCL all-reduces issued): {val('bench_collective_1rank')} grad-steps/s |",
 f"| `{RD}_bench_unfused_gn.json`, `{RD}_bench_unfused_proj.json`, `{RD}_bench_gemm_f32.json`, `{RD}_bench_trunk_f32.json`, `{RD}_bench_noise_hash.json`, `{RD}_bench_chain_unfused.json` | same-call variants of the official line (the switches that are some test's reference arithmetic; the others went with round 6): `SERL_GN_FUSE=0` {ms('bench_unfused_gn')} ms, `SERL_PROJ_FUSE=0` {ms('bench_unfused_proj')} ms, `SERL_GEMM=f32` {ms('bench_gemm_f32')} ms, `--trunk f32` {ms('bench_trunk_f32')} ms, `--noise hash` {ms('bench_noise_hash')} ms, `SERL_CHAIN_FUSE=0` {ms('bench_chain_unfused')} ms (default {b['ms_per_step']}, again {ms('bench_again')}) |",
 f"| `scaling_pieces.json` | the pieces of the two multi-GPU designs on ONE GPU, read by `bench.py` into the line's `projection`: {json.dumps(sp)} |",
 f"| `pmc_traffic.json` | HBM traffic per launch (`--pmc FETCH_SIZE` / `WRITE_SIZE` in separate passes, FETCH doubled per the gfx950 note): {json.dumps(tr)[:600]} |",
 f"| `{RD}_actor_latency.json`, `{RD}_sac_state.json` | `sample_actions` on one observation: {J.get('actor_latency', {}).get('ms_per_call', J.get('actor_latency'))}; state-only SAC: {val('sac_state')} |",
]
txt = "\n".join(rows)
print(txt)
rp = os.path.join(P, "README.md")
s = open(rp).read()
if "<!-- r06-table-begin -->" in s:
    s = re.sub(r"<!-- r06-table-begin -->.*<!-- r06-table-end -->", "<!-- r06-table-begin -->\n| file | what |\n|---|---|\n" + txt.replace("\\", "\\\\") + "\n<!-- r06-table-end -->", s, flags=re.S)
    open(rp, "w").write(s)
