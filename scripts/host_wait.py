"""From a rocprofv3 --kernel-trace --hip-runtime-trace CSV pair: for the last traced learner steps, list the host's long API calls
(> 30 us) and the first / last kernels of every trunk pass on the same clock, to see what the host was waiting for when the
trunk stream ran dry.  usage: python scripts/host_wait.py <dir>"""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ht = [f for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True)]
ht = ht[0] if ht else None
ks = []
for r in csv.DictReader(open(kt)):
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
ks.sort()
g = [i for i, k in enumerate(ks) if "gather_crop" in k[2]]
if len(g) < 4:
    print("too few steps"); sys.exit(0)
t0 = ks[g[-4]][0]
ev = []
for i in range(g[-4], len(ks)):
    s, e, n, q = ks[i]
    short = n.split("(")[0].replace("void serl::", "").replace("serl::", "")[:44]
    if any(x in n for x in ("gather_crop", "block_out_split", "adam_ema", "sle_proprio", "copyBuffer", "conv_init")):
        ev.append((s - t0, f"GPU q{q} start {short} (runs {(e - s) / 1e3:.1f} us)"))
        ev.append((e - t0, f"GPU q{q} end   {short}"))
if ht:
    for r in csv.DictReader(open(ht)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t0:
            continue
        fn = r.get("Function", r.get("Name", "?"))
        if e - s > 30000 or fn in ("hipEventSynchronize", "hipStreamSynchronize"):
            ev.append((s - t0, f"HOST call  {fn} ... {(e - s) / 1e3:.1f} us"))
            ev.append((e - t0, f"HOST ret   {fn}"))
        if fn == "hipMemcpyAsync":
            ev.append((s - t0, "HOST hipMemcpyAsync (gather parameters of the next pass are being enqueued)"))
else:
    print("no hip api trace found:", os.listdir(d))
for t, m in sorted(ev):
    print(f"{t / 1e3:10.1f} us  {m}")
