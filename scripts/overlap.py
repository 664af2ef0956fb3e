"""Which kernels overlap a given kernel (by name substring) in a rocprofv3 --kernel-trace CSV: for the LAST few instances prints
the instance's duration and the kernels of OTHER queues whose execution intersects it (with the overlap in us)."""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
pat = sys.argv[2]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void serl::", "").replace("serl::", "")[:40]
inst = [r for r in rows if pat in r["Kernel_Name"]][-3:]
for k in inst:
    s, e = int(k["Start_Timestamp"]), int(k["End_Timestamp"])
    print(f"{name(k)} queue {k['Queue_Id']}: {(e - s) / 1000:.1f} us, grid {k.get('Grid_Size', '?')} wg {k.get('Workgroup_Size', '?')}")
    tot = 0.0
    for r in rows:
        if r["Queue_Id"] == k["Queue_Id"]:
            continue
        s2, e2 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ov = min(e, e2) - max(s, s2)
        if ov > 0:
            tot += ov / 1000
            print(f"    {name(r):42s} {(e2 - s2) / 1000:7.1f} us (overlap {ov / 1000:6.1f}) grid {r.get('Grid_Size', '?')} lds {r.get('LDS_Block_Size', '?')} vgpr {r.get('VGPR_Count', '?')}")
    print(f"    total overlapped kernel time {tot:.1f} us")
