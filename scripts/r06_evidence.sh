#!/bin/bash
# Round-6 evidence, ONE gpurun call on ONE commit:
#     /usr/local/graft/bin/gpurun --timeout 3000 -- "bash scripts/r06_evidence.sh $(git rev-parse --short HEAD)$(git diff --quiet || echo -dirty)"
# The GPU box has no .git, so the commit the numbers belong to is passed in and written into every derived file (commit.txt,
# pmc_traffic.json, frac_from_stats.txt, scaling_pieces.json); bench.py reports it as roofline.traffic_source / projection.commit.
# Items: the official bench line; rocprofv3 --kernel-trace --stats of the same command (pipelined / serial / SmallEncoder); the PMC passes
# (HBM traffic, MFMA busy, LDS conflicts, waits -- every counter group in its own pass, --kernel-trace only); per-launch traces; the pieces of
# the two multi-GPU designs; every other bench line and the same-call variants.  Everything lands in gpurun_out/r06_evidence/;
# scripts/r06_refresh.py copies the summaries into profiles/r06_* and rewrites the table of profiles/README.md.
set -x
C=${1:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_evidence; rm -rf $O; mkdir -p $O
echo "$C" > $O/commit.txt
cd $R
date +%s > $O/t0
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
PB="--no-cpu-baseline --no-verify --fill 3000 --steps 30 --warmup 5 --repeats 1"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py $PB > $O/stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o s -- python $R/bench.py $PB --no-pipeline > $O/stats_serial.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats_small -o s -- python $R/bench.py $PB --steps 8 --encoder small --no-pipeline > $O/stats_small.log 2>&1
(cd $R && python scripts/rocprof_summary.py $(find $O/stats -name '*results.db' | head -1) $O/kernel_stats.csv; python scripts/rocprof_summary.py $(find $O/stats_serial -name '*results.db' | head -1) $O/kernel_stats_serial.csv; python scripts/rocprof_summary.py $(find $O/stats_small -name '*results.db' | head -1) $O/kernel_stats_small.csv)
(cd $R && { echo "# commit $C"; python scripts/frac_from_stats.py $O/kernel_stats.csv; python scripts/frac_from_stats.py $O/kernel_stats_serial.csv; } > $O/frac_from_stats.txt)
BA="--no-cpu-baseline --no-verify --no-pipeline --fill 1500 --steps 6 --warmup 2 --repeats 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py $BA > $O/pmc_$c.log 2>&1
done
(cd $R && python scripts/pmc_to_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json $C)
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py $BA > $O/pmc_mfma.log 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -o p -- python $R/bench.py $BA > $O/pmc_lds.log 2>&1
(cd $R && python scripts/pmc_counters.py $O/pmc_mfma $O/pmc_lds $O/mfma_counters.json)
timeout 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_wait -o p -- python $R/bench.py $BA > $O/pmc_wait.log 2>&1
(cd $R && python scripts/pmc_wait.py $O/pmc_wait $O/wait_counters.json)
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py $PB --steps 12 > $O/trace.log 2>&1
(cd $R && python scripts/timeline_full.py $O/trace > $O/timeline.txt 2>&1; python scripts/chain_trace.py $O/trace > $O/launches_pipelined.txt 2>&1; python scripts/timeline_streams.py $O/trace > $O/timeline_streams.txt 2>&1)
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_serial -o t -- python $R/bench.py $PB --steps 12 --no-pipeline > $O/trace_serial.log 2>&1
(cd $R && python scripts/chain_trace.py $O/trace_serial > $O/launches_serial.txt 2>&1)
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_upd -o t -- python $R/bench.py $PB --steps 12 --farm-role updater > $O/trace_upd.log 2>&1
(cd $R && python scripts/chain_trace.py $O/trace_upd > $O/launches_farm_updater.txt 2>&1)
find $O -name '*.db' -delete; find $O -name '*.csv' -size +2M -delete
cd $R
date +%s > $O/t1
NB="--no-cpu-baseline --steps 110 --repeats 3"
timeout 150 python bench.py $NB --no-pipeline > $O/bench_serial.json 2> /dev/null
for w in 2 4 8; do timeout 100 python bench.py $NB --emulate-world $w > $O/bench_emulate_world$w.json 2> /dev/null; done
timeout 100 python bench.py $NB --farm-role worker > $O/bench_farm_worker.json 2> /dev/null
timeout 100 python bench.py $NB --farm-role updater > $O/bench_farm_updater.json 2> /dev/null
timeout 100 python bench.py $NB > $O/bench_again.json 2> /dev/null
python - <<PY
import json
def ms(f):
    d = json.load(open("$O/" + f)); return d.get("ms_per_step", d.get("diagnostic_ms_per_step"))
json.dump({"commit": "$C", "one_gpu_ms": ms("bench_again.json"), "farm_worker_ms": ms("bench_farm_worker.json"), "farm_updater_ms": ms("bench_farm_updater.json"),
           "dp_share_ms": {str(w): ms(f"bench_emulate_world{w}.json") for w in (2, 4, 8)},
           "how": "scripts/r06_evidence.sh, one call on one box: bench.py --emulate-world N / --farm-role worker | updater / the default line, 3 x 110 steps each"},
          open("$O/scaling_pieces.json", "w"), indent=1)
PY
timeout 200 python bench.py --no-cpu-baseline --steps 40 --repeats 3 --encoder small > $O/bench_small_encoder.json 2> /dev/null
for w in drq_demos peg fwbw; do timeout 150 python bench.py --workload $w --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline > $O/bench_$w.json 2> /dev/null; done
timeout 100 python bench.py $NB --car 4 --steps 50 > $O/bench_car4.json 2> /dev/null
timeout 100 python bench.py $NB --force-collective 2> /dev/null | tail -1 > $O/bench_collective_1rank.json
timeout 100 python bench.py $NB --emulate-world 8 --force-collective 2> /dev/null | tail -1 > $O/bench_emulate_world8_collective.json
timeout 100 python bench.py $NB --trunk f32 --steps 40 > $O/bench_trunk_f32.json 2> /dev/null
SERL_GN_FUSE=0 timeout 100 python bench.py $NB > $O/bench_unfused_gn.json 2> /dev/null
SERL_PROJ_FUSE=0 timeout 100 python bench.py $NB > $O/bench_unfused_proj.json 2> /dev/null
SERL_GEMM=f32 timeout 100 python bench.py $NB > $O/bench_gemm_f32.json 2> /dev/null
timeout 100 python bench.py $NB --noise hash > $O/bench_noise_hash.json 2> /dev/null
SERL_CHAIN_FUSE=0 timeout 100 python bench.py $NB > $O/bench_chain_unfused.json 2> /dev/null
timeout 100 python bench.py --workload actor_latency > $O/actor_latency.json 2> /dev/null
timeout 150 python bench.py --workload sac_state --steps 200 > $O/sac_state.json 2> /dev/null
date +%s > $O/t2
echo "profiling part $(( $(cat $O/t1) - $(cat $O/t0) )) s, bench lines $(( $(cat $O/t2) - $(cat $O/t1) )) s"
head -c 600 $O/bench.json; echo; cat $O/frac_from_stats.txt; cat $O/scaling_pieces.json
