"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; scripts/r06_evidence.sh).

usage: python scripts/pmc_to_json.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [commit the passes ran on]

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB; on gfx950
FETCH_SIZE counts 64 B per 128-B request, so it is doubled.  "per launch" = mean over the dispatches of the
kernel family in the pass (the trunk's 11 conv launches have different shapes: the mean matches bench.py's
flop_per_launch_avg convention)."""
import collections, csv, glob, json, os, sys


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return tot, disp


def family(tot, disp, pats):
    ks = [k for k in tot if any(p in k for p in pats)]
    n = sum(len(disp[k]) for k in ks)
    return (sum(tot[k] for k in ks) / n if n else 0.0), n


ft, fd = load(sys.argv[1])
wt, wd = load(sys.argv[2])
CONV16 = ("conv_igemm_f16x3_kernel", "conv3x3_rowslab_f16x3_kernel", "conv3x3_slabdma_f16x3_kernel", "conv_dma_f16x3_kernel")
out = {"commit": sys.argv[4] if len(sys.argv) > 4 else None,
       "units": "bytes; FETCH_SIZE/WRITE_SIZE are KiB counters, FETCH_SIZE doubled (gfx950 counts 64 B per 128-B "
                "request: MI355X_MICROARCH.md HBM section); mean over the dispatches of the family"}
for name, pats in (("conv_igemm_f16x3", CONV16), ("conv_igemm_f32", ("conv_igemm_kernel",)), ("gather_crop", ("gather_crop_kernel", "gather_crop_rgb_kernel")),
                   ("conv_init_f16x3", ("conv_init_f16x3_kernel", "conv_init_u8_kernel")), ("gn_relu_maxpool", ("gn_relu_maxpool", "pool_finish_split")),
                   ("block_out", ("block_out_split",)), ("gn_relu_split", ("gn_relu_split_kernel",)), ("adam_ema", ("adam_ema",)),
                   ("gemm_f32", ("gemm_f32_kernel",)), ("gemm_bf16x3", ("gemm_bf16x3_kernel",)), ("zero_sys", ("zero_sys_kernel",))):
    f, nf = family(ft, fd, pats)
    w, nw = family(wt, wd, pats)
    if nf == 0 and nw == 0:
        continue
    fb, wb = 2.0 * f * 1024.0, w * 1024.0
    out[f"{name}_bytes_per_launch"] = int(fb + wb)
    out[f"{name}_fetch_bytes_per_launch"] = int(fb)
    out[f"{name}_write_bytes_per_launch"] = int(wb)
    out[f"{name}_dispatches_in_pass"] = nf
raw = {}
for k in set(ft) | set(wt):
    raw[k[:80]] = {"FETCH_SIZE_KiB": round(ft.get(k, 0.0) / max(1, len(fd.get(k, ()))), 1),
                   "WRITE_SIZE_KiB": round(wt.get(k, 0.0) / max(1, len(wd.get(k, ()))), 1)}
out["raw_KiB_per_dispatch"] = dict(sorted(raw.items(), key=lambda kv: -(kv[1]["FETCH_SIZE_KiB"] + kv[1]["WRITE_SIZE_KiB"])))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print({k: v for k, v in out.items() if k.endswith("_bytes_per_launch")})
