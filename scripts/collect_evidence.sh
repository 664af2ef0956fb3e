#!/bin/bash
# Run on the GPU box (via gpurun): official bench line and its variants, the BASELINE configs 2-4, rocprofv3 kernel stats, PMC
# passes (HBM traffic, MFMA busy, LDS bank conflicts -- each counter group in its own pass, --kernel-trace only), timeline.
# Outputs land in gpurun_out/evidence/ ; scripts/refresh_profiles.py copies the summaries into profiles/ as r04_*.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence; rm -rf $O; mkdir -p $O
cd $R
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err
NB="--no-cpu-baseline --steps 110 --repeats 3"   # (>= 320 timed steps: bench.py then times every 16th launch, like the official line)
timeout 200 python bench.py $NB --no-pipeline > $O/bench_serial.json 2> $O/bench_serial.err
timeout 200 python bench.py $NB --trunk f32 --steps 40 > $O/bench_f32.json 2> $O/bench_f32.err
timeout 200 python bench.py $NB --car 4 --steps 50 > $O/bench_car4.json 2> $O/bench_car4.err
SERL_GN_FUSE=0 timeout 200 python bench.py $NB > $O/bench_unfused_gn.json 2> /dev/null
SERL_GEMM=f32 timeout 200 python bench.py $NB > $O/bench_gemm_f32.json 2> /dev/null
for w in 2 4 8; do timeout 200 python bench.py $NB --emulate-world $w > $O/bench_emulate_world$w.json 2> $O/bench_emu$w.err; done
for w in drq_demos peg fwbw; do timeout 300 python bench.py --workload $w --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python bench.py --no-cpu-baseline --steps 40 --repeats 3 --encoder small > $O/bench_small_encoder.json 2> $O/bench_small.err
timeout 200 python bench.py $NB --force-collective > $O/bench_collective_1rank.json 2> /dev/null
timeout 200 python bench.py $NB --emulate-world 8 --force-collective > $O/bench_emulate_world8_collective.json 2> /dev/null
# update-chain variants, same call: one launch per operation (round-3 schedule) and the opt-in LayerNorm epilogues
for m in "" "--emulate-world 8" "--no-pipeline"; do
  t=$(echo $m | tr -d ' -')
  SERL_CHAIN_FUSE=0 timeout 200 python bench.py $NB $m > $O/bench_chain_unfused$t.json 2> /dev/null
  SERL_CHAIN_LN_EPI=1 timeout 200 python bench.py $NB $m > $O/bench_chain_lnepi$t.json 2> /dev/null
done
timeout 200 python bench.py --workload actor_latency > $O/actor_latency.json 2> /dev/null
timeout 300 python bench.py --workload sac_state --steps 200 > $O/sac_state.json 2> /dev/null
timeout 200 python scripts/probes/replay_race.py 1500 48 > $O/replay_race.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PB="--no-cpu-baseline --no-verify --fill 3000 --steps 30 --warmup 5 --repeats 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py $PB > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o s -- python $R/bench.py $PB --no-pipeline > $O/stats_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_small -o s -- python $R/bench.py $PB --steps 8 --encoder small --no-pipeline > $O/stats_small.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py $PB --steps 12 > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_serial -o t -- python $R/bench.py $PB --steps 12 --no-pipeline > $O/trace_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_e8 -o t -- python $R/bench.py $PB --steps 12 --emulate-world 8 > $O/trace_e8.log 2>&1
BA="--no-cpu-baseline --no-verify --no-pipeline --fill 1500 --steps 6 --warmup 2 --repeats 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py $BA > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py $BA > $O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -o p -- python $R/bench.py $BA > $O/pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_wait -o p -- python $R/bench.py $BA > $O/pmc_wait.log 2>&1
cd $R
python scripts/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json
python scripts/pmc_counters.py $O/pmc_mfma $O/pmc_lds $O/mfma_counters.json
python scripts/pmc_wait.py $O/pmc_wait $O/wait_counters.json
for t in stats stats_serial stats_small; do python scripts/rocprof_summary.py $(find $O/$t -name '*results.db' | head -1) $O/kernel_$t.csv || true; done
python scripts/frac_from_stats.py $O/kernel_stats.csv > $O/frac_from_stats.txt; python scripts/frac_from_stats.py $O/kernel_stats_serial.csv >> $O/frac_from_stats.txt; cat $O/frac_from_stats.txt
python scripts/timeline_full.py $O/trace > $O/timeline.txt 2>&1
python scripts/chain_trace.py $O/trace > $O/launches_pipelined.txt 2>&1
python scripts/chain_trace.py $O/trace_serial > $O/launches_serial.txt 2>&1
python scripts/chain_trace.py $O/trace_e8 > $O/launches_emulate_world8.txt 2>&1
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
ls -R $O | head -80
