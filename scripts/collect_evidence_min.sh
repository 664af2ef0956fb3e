#!/bin/bash
# Short evidence call (when the round's GPU budget no longer covers scripts/collect_evidence.sh): the official bench line, the
# rocprofv3 kernel statistics of the same command, the SmallEncoder line, the serial schedule and one rank's share of 8, then the
# GPU test suite in whatever time is left (most important first; every step has its own timeout and writes its own file).
# Outputs land in gpurun_out/evidence_min/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence_min; rm -rf $O; mkdir -p $O
cd $R
date +%s > $O/t0
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
PB="--no-cpu-baseline --no-verify --fill 3000 --steps 30 --warmup 5 --repeats 1"
(cd /tmp && TMPDIR=/tmp timeout 200 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py $PB > $O/stats.log 2>&1)
python scripts/rocprof_summary.py $(find $O/stats -name '*results.db' | head -1) $O/kernel_stats.csv
python scripts/frac_from_stats.py $O/kernel_stats.csv > $O/frac_from_stats.txt
timeout 200 python bench.py --no-cpu-baseline --steps 40 --repeats 3 --encoder small > $O/bench_small_encoder.json 2> $O/bench_small.err
NB="--no-cpu-baseline --steps 110 --repeats 3"
timeout 150 python bench.py $NB --no-pipeline > $O/bench_serial.json 2> $O/bench_serial.err
timeout 150 python bench.py $NB --emulate-world 8 > $O/bench_emulate_world8.json 2> $O/bench_emu8.err
date +%s > $O/t1
# the suite last: it is the longest item and the driver repeats it at round end anyway
# (files whose kernels changed last go first; -v so that a run cut short still says how far it got)
FIRST="tests/test_small_encoder_gpu.py tests/test_golden_update_gpu.py tests/test_agent_gpu.py tests/test_timed_shapes_gpu.py tests/test_bench_shape_gpu.py tests/test_chain_fusion_gpu.py"
REST=$(ls tests/test_*_gpu.py tests/test_abi.py | grep -v -e small_encoder -e golden_update -e test_agent_gpu -e timed_shapes -e bench_shape -e chain_fusion | tr '\n' ' ')
timeout ${SUITE_TIMEOUT:-330} python -m pytest $FIRST $REST -m gpu -x -v --durations=25 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
date +%s > $O/t2
find $O -name '*.db' -delete; find $O -name '*.csv' -size +2M -delete
tail -5 $O/pytest_gpu.log; cat $O/bench.json | cut -c1-400
