"""Recomputes bench.py's roofline.frac for the block-conv family from a rocprofv3 --kernel-trace --stats summary
(profiles/r02_kernel_stats*.csv written by scripts/rocprof_summary.py), independently of the HIP events inside bench.py:
  conv time per trunk pass = sum of total_us over the block-conv kernels / number of trunk passes (= conv_init calls)
  frac = 3 x algorithmic FLOPs of the 11 block convs per pass / that time / 2.5 PFLOP/s.
usage: python scripts/frac_from_stats.py <kernel_stats.csv> [images per pass = 1024]"""
import csv, sys
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sys.path.insert(0, ".")
import bench
macs = bench.conv_macs_per_image()
flop_pass = sum(2.0 * m * n_img for t, m in macs.items() if t.startswith("conv_igemm"))
conv_us, passes = 0.0, 0
for r in csv.DictReader(open(sys.argv[1])):
    k = r["kernel"]
    if any(p in k for p in ("conv_dma_f16x3_kernel", "conv3x3_rowslab_f16x3_kernel", "conv3x3_slabdma_f16x3_kernel", "conv_igemm_f16x3_kernel")):
        conv_us += float(r["total_us"])
    if "conv_init_u8_kernel" in k and "pack" not in k:   # one conv_init launch per trunk pass (NOT the one-off weight packing kernel)
        passes += int(r["calls"])
t = conv_us / passes
print(f"{passes} trunk passes, block convs {t:.1f} us per pass, {flop_pass/1e9:.1f} algorithmic GFLOP per pass -> "
      f"{flop_pass/t/1e6:.1f} algorithmic TFLOP/s, executed x3 = {3*flop_pass/t/1e6:.1f} TFLOP/s, frac {3*flop_pass/t/1e6/2500.0:.4f}")
