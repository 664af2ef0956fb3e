"""Aggregate a rocprofv3 --pmc CSV (counter_collection) per kernel: mean counter value per dispatch."""
import csv, sys, collections, glob, os
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    if pat and pat not in k:
        continue
    n = len(cnt[k])
    print(k, "dispatches", n, {c: round(v / n, 1) for c, v in acc[k].items()})
