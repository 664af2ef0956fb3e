"""Rewrites the measured numbers of DESIGN.md (section 4 'Measured on ...' paragraph, step anatomy, section 5 table)
from profiles/*.json.  Run after scripts/refresh_profiles.py."""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = lambda n: json.load(open(os.path.join(R, "profiles", n)))
b, f32, se = L("r01_bench.json"), L("r01_bench_trunk_f32.json"), L("r01_bench_serial.json")
e2, e4, e8 = L("r01_bench_emulate_world2.json"), L("r01_bench_emulate_world4.json"), L("r01_bench_emulate_world8.json")
pm = L("pmc_traffic.json")
r, cb = b["roofline"], b["cpu_baseline"]
p = os.path.join(R, "DESIGN.md")
s = open(p).read()
m0, m1 = s.index("Measured on 1× MI355X (see `profiles/README.md` for the full log):"), s.index("**Why the convs stop at")
s = s[:m0] + f"""Measured on 1× MI355X (see `profiles/README.md` for the full log): **{b['value']} grad-steps/s**
({b['ms_per_step']} ms/step, CAR=1; box-to-box variation ±3 %), conv family {r['algorithmic_tflops']} algorithmic TFLOP/s
({r['achieved']} TFLOP/s executed = {100*r['frac']:.1f} % of the fp16-MFMA peak while co-running with the update chain;
{se['roofline']['achieved']} TFLOP/s = {100*se['roofline']['frac']:.1f} % alone);
exact-fp32 trunk: {f32['value']} grad-steps/s, conv family {f32['roofline']['achieved']} TFLOP/s = {100*f32['roofline']['frac']:.1f} % of the fp32-MFMA peak;
sample+aug kernel {se['roofline']['sample_aug_hbm']['achieved']} TB/s alone / {r['sample_aug_hbm']['achieved']} TB/s co-running ({pm['gather_crop_bytes_per_launch']/1e6:.1f} MB of HBM traffic for 100.72 MB algorithmic); CPU port
(`cpu_baseline`, {cb['cores']} usable cores, PyTorch-CPU fp32, 2 trunk passes) {cb['value']} grad-steps/s → {b['value']/cb['value']:.0f}×.

""" + s[m1:]
a0, a1 = s.index("**Step anatomy (B=256).**"), s.index("## 5. Multi-GPU")
pk = se["roofline"]["per_kernel"]
conv = sum(v["avg_us"] for k, v in pk.items() if k.startswith("conv_igemm")) / 1e3
s = s[:a0] + f"""**Step anatomy (B=256).**  Serial: trunk ≈3.1 ms (convs {conv:.2f}, conv_init {pk['conv_init']['avg_us']/1e3:.2f}, pool finish {pk['gn_relu_maxpool']['avg_us']/1e3:.2f}, block_out ≈0.27,
gn_relu_split ≈0.18) + update chain ≈0.9 ms = {se['ms_per_step']} ms.  Pipelined: {b['ms_per_step']} ms — the chain's kernels
can only start in wave slots that retiring conv workgroups free, which is why making them small (lean GEMM) and few
(multi-instance launches) moved the single-GPU number although the chain is not the longer of the two streams.

""" + s[a1:]
t0, t1 = s.index("| N | per-rank batch | ms/step | grad-steps/s | × 1 GPU |"), s.index("Strong scaling of a 3.5 ms step is latency-bound")
s = s[:t0] + f"""| N | per-rank batch | ms/step | grad-steps/s | × 1 GPU |
|---|---|---|---|---|
| 1 | 256 | {b['ms_per_step']} | {b['value']} | 1.00 |
| 2 | 128 | {e2['ms_per_step']} | {e2['value']} | {e2['value']/b['value']:.2f} |
| 4 | 64 | {e4['ms_per_step']} | {e4['value']} | {e4['value']/b['value']:.2f} |
| 8 | 32 | {e8['ms_per_step']} | {e8['value']} | {e8['value']/b['value']:.2f} |

""" + s[t1:]
open(p, "w").write(s)
print("DESIGN.md refreshed")
