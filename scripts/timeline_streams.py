"""Per-queue view of one steady-state step from a rocprofv3 --kernel-trace CSV: for each HIP stream
(hardware queue) the busy time, idle gaps and the largest gaps with the kernels either side."""
import csv, sys, glob, os, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gather_crop" in r["Kernel_Name"]]
a, b = idx[-4], idx[-2]          # two steps
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
print(f"2 steps wall {(t1-t0)/1000:.1f} us ({(t1-t0)/2000:.1f} per step), {len(seg)} kernels")
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void serl::", "").replace("serl::", "")[:34]
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = []
    for p, n in zip(rs, rs[1:]):
        gaps.append((int(n["Start_Timestamp"]) - int(p["End_Timestamp"]), name(p), name(n)))
    print(f"queue {q}: {len(rs)} kernels, busy {busy/1000:.1f} us, span {(int(rs[-1]['End_Timestamp'])-int(rs[0]['Start_Timestamp']))/1000:.1f} us, "
          f"sum gaps {sum(g[0] for g in gaps)/1000:.1f} us")
    for g in sorted(gaps, reverse=True)[:8]:
        print(f"    gap {g[0]/1000:7.1f} us  {g[1]} -> {g[2]}")
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in rs:
        per[name(r)][0] += 1; per[name(r)][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"      {k:36s} n={n:3d} total={us:8.1f} avg={us/n:6.1f}")
