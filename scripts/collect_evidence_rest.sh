#!/bin/bash
# Second short evidence call on the final code (after scripts/collect_evidence_min.sh): PMC passes (HBM traffic, MFMA busy, LDS
# conflicts, waits -- each counter group in its own pass, --kernel-trace only), the serial / SmallEncoder kernel statistics, the
# per-launch traces, then the remaining bench lines.  Every item writes its own file and is summarised right away, so a call
# that runs out of budget keeps what it finished.  Outputs land in gpurun_out/evidence_rest/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence_rest; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-verify --no-pipeline --fill 1500 --steps 6 --warmup 2 --repeats 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py $BA > $O/pmc_$c.log 2>&1
done
(cd $R && python scripts/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json)
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py $BA > $O/pmc_mfma.log 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -o p -- python $R/bench.py $BA > $O/pmc_lds.log 2>&1
(cd $R && python scripts/pmc_counters.py $O/pmc_mfma $O/pmc_lds $O/mfma_counters.json)
timeout 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_wait -o p -- python $R/bench.py $BA > $O/pmc_wait.log 2>&1
(cd $R && python scripts/pmc_wait.py $O/pmc_wait $O/wait_counters.json)
# the SmallEncoder path's counters
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma_small -o p -- python $R/bench.py $BA --encoder small --steps 3 > $O/pmc_mfma_small.log 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds_small -o p -- python $R/bench.py $BA --encoder small --steps 3 > $O/pmc_lds_small.log 2>&1
(cd $R && python scripts/pmc_counters.py $O/pmc_mfma_small $O/pmc_lds_small $O/mfma_counters_small_encoder.json)
PB="--no-cpu-baseline --no-verify --fill 3000 --steps 30 --warmup 5 --repeats 1"
timeout 120 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o s -- python $R/bench.py $PB --no-pipeline > $O/stats_serial.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $O/stats_small -o s -- python $R/bench.py $PB --steps 8 --encoder small --no-pipeline > $O/stats_small.log 2>&1
for t in stats_serial stats_small; do (cd $R && python scripts/rocprof_summary.py $(find $O/$t -name '*results.db' | head -1) $O/kernel_$t.csv); done
(cd $R && python scripts/frac_from_stats.py $O/kernel_stats_serial.csv > $O/frac_from_stats_serial.txt)
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py $PB --steps 12 > $O/trace.log 2>&1
(cd $R && python scripts/timeline_full.py $O/trace > $O/timeline.txt 2>&1; python scripts/chain_trace.py $O/trace > $O/launches_pipelined.txt 2>&1)
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_serial -o t -- python $R/bench.py $PB --steps 12 --no-pipeline > $O/trace_serial.log 2>&1
(cd $R && python scripts/chain_trace.py $O/trace_serial > $O/launches_serial.txt 2>&1)
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_e8 -o t -- python $R/bench.py $PB --steps 12 --emulate-world 8 > $O/trace_e8.log 2>&1
(cd $R && python scripts/chain_trace.py $O/trace_e8 > $O/launches_emulate_world8.txt 2>&1)
find $O -name '*.db' -delete; find $O -name '*.csv' -size +2M -delete
cd $R
NB="--no-cpu-baseline --steps 110 --repeats 3"
for w in 2 4; do timeout 100 python bench.py $NB --emulate-world $w > $O/bench_emulate_world$w.json 2> /dev/null; done
for w in drq_demos peg fwbw; do timeout 150 python bench.py --workload $w --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline > $O/bench_$w.json 2> /dev/null; done
timeout 100 python bench.py $NB --car 4 --steps 50 > $O/bench_car4.json 2> /dev/null
timeout 100 python bench.py $NB --force-collective > $O/bench_collective_1rank.json 2> /dev/null
timeout 100 python bench.py $NB --emulate-world 8 --force-collective > $O/bench_emulate_world8_collective.json 2> /dev/null
timeout 100 python bench.py $NB --trunk f32 --steps 40 > $O/bench_f32.json 2> /dev/null
SERL_GN_FUSE=0 timeout 100 python bench.py $NB > $O/bench_unfused_gn.json 2> /dev/null
SERL_GEMM=f32 timeout 100 python bench.py $NB > $O/bench_gemm_f32.json 2> /dev/null
timeout 100 python bench.py --workload actor_latency > $O/actor_latency.json 2> /dev/null
timeout 150 python bench.py --workload sac_state --steps 200 > $O/sac_state.json 2> /dev/null
for m in "" "--emulate-world 8" "--no-pipeline"; do
  t=$(echo $m | tr -d ' -')
  SERL_CHAIN_FUSE=0 timeout 100 python bench.py $NB $m > $O/bench_chain_unfused$t.json 2> /dev/null
  SERL_CHAIN_LN_EPI=1 timeout 100 python bench.py $NB $m > $O/bench_chain_lnepi$t.json 2> /dev/null
done
ls $O
