"""Where the waves' time goes, per kernel family, from one rocprofv3 --pmc pass with
SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES (serial schedule):
  wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES: share of a wave's lifetime parked on s_waitcnt / barriers (both quad-cycle units)
usage: python scripts/pmc_wait.py <pass dir> <out.json>"""
import collections, csv, glob, json, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
FAM = (("conv_dma_f16x3", "conv_dma_f16x3_kernel"), ("conv3x3_rowslab_f16x3", "conv3x3_rowslab_f16x3_kernel"),
       ("conv3x3_slabdma_f16x3", "conv3x3_slabdma_f16x3_kernel"),
       ("conv_init_u8", "conv_init_u8_kernel<"), ("gemm_bf16x3", "gemm_bf16x3_kernel"), ("pool_finish_split", "pool_finish_split_kernel"))
out = {"note": __doc__.strip()}
for name, pat in FAM:
    ks = [k for k in tot if pat in k]
    n = sum(len(disp[k]) for k in ks)
    if not n:
        continue
    cs = collections.defaultdict(float)
    for k in ks:
        for c, v in tot[k].items():
            cs[c] += v
    rec = {"dispatches": n}
    rec.update({c + "_per_dispatch": round(v / n, 1) for c, v in cs.items()})
    if cs.get("SQ_WAVE_CYCLES"):
        rec["wait_any_frac"] = round(cs.get("SQ_WAIT_ANY", 0.0) / cs["SQ_WAVE_CYCLES"], 4)
        rec["wait_inst_lds_frac"] = round(cs.get("SQ_WAIT_INST_LDS", 0.0) / cs["SQ_WAVE_CYCLES"], 4)
    out[name] = rec
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "frac" in kk} for k, v in out.items() if isinstance(v, dict)}))
