#!/bin/bash
# Run on the GPU box (via gpurun): official bench line and its variants, rocprofv3 kernel stats, PMC passes (HBM traffic,
# MFMA busy, LDS bank conflicts -- each counter group in its own pass, --kernel-trace only).
# Outputs land in gpurun_out/evidence/ ; scripts/refresh_profiles.py copies the summaries into profiles/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --no-cpu-baseline --no-pipeline > $O/bench_serial.json 2> $O/bench_serial.err
timeout 200 python bench.py --trunk f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err
timeout 200 python bench.py --car 4 --no-cpu-baseline > $O/bench_car4.json 2> $O/bench_car4.err
SERL_GN_FUSE=0 timeout 200 python bench.py --no-cpu-baseline > $O/bench_unfused_gn.json 2> /dev/null
for w in 2 4 8; do timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world $w > $O/bench_emulate_world$w.json 2> $O/bench_emu$w.err; done
timeout 200 python bench.py --workload actor_latency > $O/actor_latency.json 2> /dev/null
timeout 300 python bench.py --workload sac_state --steps 200 > $O/sac_state.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py --no-cpu-baseline --fill 3000 --steps 30 --warmup 5 > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o s -- python $R/bench.py --no-cpu-baseline --no-pipeline --fill 3000 --steps 30 --warmup 5 > $O/stats_serial.log 2>&1
BA="--no-cpu-baseline --no-pipeline --fill 1500 --steps 6 --warmup 2"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py $BA > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py $BA > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -o p -- python $R/bench.py $BA > $O/pmc_lds.log 2>&1
cd $R
python scripts/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json
python scripts/pmc_counters.py $O/pmc_mfma $O/pmc_lds $O/mfma_counters.json
python scripts/rocprof_summary.py $(find $O/stats -name '*results.db' | head -1) $O/kernel_stats.csv || true
python scripts/rocprof_summary.py $(find $O/stats_serial -name '*results.db' | head -1) $O/kernel_stats_serial.csv || true
python scripts/frac_from_stats.py $O/kernel_stats.csv > $O/frac_from_stats.txt; python scripts/frac_from_stats.py $O/kernel_stats_serial.csv >> $O/frac_from_stats.txt; cat $O/frac_from_stats.txt
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
ls -R $O | head -60
