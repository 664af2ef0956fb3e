#!/bin/bash
# Run on the GPU box (via gpurun): official bench line, rocprofv3 kernel stats and PMC HBM-traffic passes.
# Outputs land in gpurun_out/evidence/ ; copy the summaries into profiles/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_f16x3.json 2> $O/bench_f16x3.err
timeout 200 python bench.py --trunk f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
timeout 200 python bench.py --car 4 --no-cpu-baseline > $O/bench_f16x3_car4.json 2> $O/bench_car4.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --fill 3000 --steps 30 --warmup 5 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-pipeline --fill 1500 --steps 6 --warmup 2 > $O/pmc_$c.log 2>&1
done
cd $R
python scripts/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json
python scripts/rocprof_summary.py $(find $O/stats -name '*results.db' | head -1) $O/kernel_stats.csv || true
# multi-GPU projection: one rank's share of a world-size-N job on this GPU (no collective)
for w in 2 4 8; do timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world $w > $O/bench_emulate_world$w.json 2> $O/bench_emu$w.err; done
timeout 200 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_serial.json 2> $O/bench_serial.err
find $O -name '*.csv' -size +8M -delete
ls -R $O | head -40
