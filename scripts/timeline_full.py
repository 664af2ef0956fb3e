"""Full two-queue timeline of ONE steady-state learner step from a rocprofv3 --kernel-trace CSV: every kernel between two
consecutive gather_crop launches with its queue, start offset and duration (us), plus per-queue busy time.
usage: python scripts/timeline_full.py <dir with *kernel_trace.csv> > profiles/rNN_timeline.txt"""
import collections
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gather_crop" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
# kernels of the other queue that started shortly before this step's gather belong to the picture too
seg = [r for r in rows if t0 - 50_000 <= int(r["Start_Timestamp"]) < t1]
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void serl::", "").replace("serl::", "")[:44]
print(f"one step: {(t1 - t0) / 1000:.1f} us between consecutive gather_crop launches, {len(seg)} kernels shown")
qs = sorted({r["Queue_Id"] for r in seg})
busy = collections.defaultdict(float)
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[r["Queue_Id"]] += (e - s) / 1000
    col = qs.index(r["Queue_Id"])
    print(f"{(s - t0) / 1000:9.1f} {'':{col * 8}s}q{r['Queue_Id']:<3s} {(e - s) / 1000:8.1f}  {name(r)}  grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')} lds={r.get('LDS_Block_Size', '?')} vgpr={r.get('VGPR_Count', '?')}")
for q in qs:
    print(f"queue {q}: busy {busy[q]:.1f} us")
