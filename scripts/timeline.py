"""Kernel timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: busy time vs gaps."""
import csv, sys, glob, os, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# find steps by the gather_crop kernel
idx = [i for i, r in enumerate(rows) if "gather_crop" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = 0; last_end = t0; gaps = []
per = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    gaps.append(max(0, s - last_end)); last_end = max(last_end, e)
    k = r["Kernel_Name"].split("(")[0].replace("void serl::", "").replace("serl::", "")[:40]
    per[k][0] += 1; per[k][1] += (e - s) / 1000
print(f"step wall {(t1-t0)/1000:.1f} us, {len(seg)} kernels, busy {busy/1000:.1f} us, gaps {sum(gaps)/1000:.1f} us (avg {sum(gaps)/len(gaps)/1000:.2f})")
for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:42s} n={n:3d} total={us:8.1f} avg={us/n:6.1f}")
