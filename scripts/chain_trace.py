"""One steady-state step of a rocprofv3 --kernel-trace CSV, in launch order: short kernel name, grid (workgroups), duration and the
gap to the previous kernel's end -- the per-launch view of the update chain (serial schedule: everything on one queue).
usage: python scripts/chain_trace.py <dir with *_kernel_trace.csv> [step index from the end, default 3]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "gather_crop" in r["Kernel_Name"]]
a, b = starts[-back - 1], starts[-back]
prev_end = {}
tot = 0.0
busy = {}
for r in rows[a:b]:
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name).replace("serl::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    q = r["Queue_Id"]
    gap = 0.0 if q not in prev_end else (s - prev_end[q]) / 1e3     # idle time of THIS queue before the kernel
    prev_end[q] = max(e, prev_end.get(q, 0))
    tot += (e - s) / 1e3
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    print(f"q{r['Queue_Id']} {name[:58]:58s} wgs={wg:6d} vgpr={r['VGPR_Count']:>4s} lds={r['LDS_Block_Size']:>6s} dur={(e - s) / 1e3:8.1f} gap={gap:7.1f}")
print(f"step: {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us wall, {tot:.1f} us of kernel time, {b - a} kernels; "
      + ", ".join(f"queue {q}: {v:.1f} us busy" for q, v in sorted(busy.items())))
