"""profiles/r0N_mfma_counters.json from two rocprofv3 --pmc passes (scripts/r06_evidence.sh):
  pass 1: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES      pass 2: SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE,
  SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY.
usage: python scripts/pmc_counters.py <dir pass 1> <dir pass 2> <out.json>

Per kernel family: the mean counter value per dispatch and two ratios --
  mfma_busy_over_sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES   (the north star's "MFMA utilisation on the encoder GEMMs":
      cycles in which a SIMD's matrix pipe is busy over cycles in which the shader engine's SQ has waves; both are summed over the
      units rocprofv3 aggregates, so the ratio -- not the absolute values -- is the figure of merit; the guide's note on
      SQ_BUSY_CYCLES per-SE accounting applies)
  mfma_util_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) (SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per 32x32x16 MFMA
      over all waves = busy cycles summed over the 1024 SIMDs; SQ_BUSY_CYCLES is summed over the 8 XCDs x 4 shader engines, so
      SQ_BUSY_CYCLES / 32 is the kernel's duration in shader cycles: checked against the kernel trace of the same pass.  The
      ratio is the fraction of cycles a SIMD's matrix pipe is busy AT THE CLOCK THE PASS RAN AT (profiled passes run below the
      2.4 GHz the 2.5 PF peak assumes, so it sits above the time-derived roofline.frac by that clock ratio)
  mfma_busy_over_wave_cycles = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES)   (SQ_WAVE_CYCLES counts quad-cycles: share of a
      wave's lifetime the matrix pipe works for it; x resident waves per SIMD = the utilisation above)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE         (share of LDS cycles lost to bank conflicts)."""
import collections, csv, glob, json, os, sys


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return tot, disp


FAMILIES = (("conv_dma_f16x3", ("conv_dma_f16x3_kernel",)), ("conv3x3_rowslab_f16x3", ("conv3x3_rowslab_f16x3_kernel",)),
            ("conv3x3_slabdma_f16x3", ("conv3x3_slabdma_f16x3_kernel",)),
            ("conv_igemm_f16x3", ("conv_igemm_f16x3_kernel",)), ("conv_init_u8", ("conv_init_u8_kernel",)),
            ("gemm_f32", ("gemm_f32_kernel",)), ("gemm_bf16x3", ("gemm_bf16x3_kernel",)), ("gather_crop_rgb", ("gather_crop_rgb_kernel",)),
            # SmallEncoder passes (--encoder small): layer 0 on the u8 frames, the input-gradient scatter
            ("small_conv0_fwd", ("small_conv0_fwd_kernel",)), ("small_conv0_wgrad", ("small_conv0_wgrad_kernel",)),
            ("small_col2im", ("small_col2im_kernel",)))
out = {"note": __doc__.split("Per kernel family:")[1].strip()}
passes = [load(sys.argv[1]), load(sys.argv[2])]
for name, pats in FAMILIES:
    rec = {}
    for tot, disp in passes:
        ks = [k for k in tot if any(p in k for p in pats)]
        n = sum(len(disp[k]) for k in ks)
        if not n:
            continue
        rec["dispatches"] = n
        cs = collections.defaultdict(float)
        for k in ks:
            for c, v in tot[k].items():
                cs[c] += v
        for c, v in cs.items():
            rec[c + "_per_dispatch"] = round(v / n, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch" in rec and rec.get("SQ_BUSY_CYCLES_per_dispatch"):
        rec["mfma_busy_over_sq_busy"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"] / rec["SQ_BUSY_CYCLES_per_dispatch"], 4)
        rec["mfma_util_per_simd"] = round(rec["mfma_busy_over_sq_busy"] / 32.0, 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch" in rec and rec.get("SQ_WAVE_CYCLES_per_dispatch"):
        rec["mfma_busy_over_wave_cycles"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"] / (4.0 * rec["SQ_WAVE_CYCLES_per_dispatch"]), 4)
    if "SQ_LDS_BANK_CONFLICT_per_dispatch" in rec and rec.get("SQ_LDS_IDX_ACTIVE_per_dispatch"):
        rec["lds_conflict_frac"] = round(rec["SQ_LDS_BANK_CONFLICT_per_dispatch"] / rec["SQ_LDS_IDX_ACTIVE_per_dispatch"], 4)
    if rec:
        out[name] = rec
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if "over" in kk or "frac" in kk or "util" in kk} for k, v in out.items() if isinstance(v, dict)}))
