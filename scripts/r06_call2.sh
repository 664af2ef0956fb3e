#!/bin/bash
# Round 6, GPU call 2: the chain's new kernels (full-row Dense+LN+tanh layer, 128x128-tile GEMM, un-drained prefetch) -- parity tests, then same-call A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call2; rm -rf $O; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests/test_agent_gpu.py tests/test_chain_fusion_gpu.py tests/test_sac_state_gpu.py tests/test_small_encoder_gpu.py tests/test_golden_update_gpu.py tests/test_classifier_gpu.py tests/test_drq_agent_gpu.py "tests/test_bench_shape_gpu.py::test_update_high_utd_at_bench_shape" "tests/test_bench_shape_gpu.py::test_update_critics_at_bench_shape" -m gpu -q -x ) > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log | cut -c1-300
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); print('$tag', d.get('ms_per_step', d.get('diagnostic_ms_per_step')), d['ms_per_step_runs'])
except Exception as e: print('$tag FAILED', open('$O/$tag.err').read()[-600:])"; }
for mode in "new" "norows SERL_DENSE_ROWS=0" "nobig SERL_GEMM_BIG=0" "neither SERL_DENSE_ROWS=0 SERL_GEMM_BIG=0"; do
  set -- $mode; m=$1; shift
  for e in "$@"; do export $e; done
  run upd_$m --farm-role updater
  run serial_$m --no-pipeline
  run emu8_$m --emulate-world 8
  run pipe_$m
  unset SERL_DENSE_ROWS SERL_GEMM_BIG
done
run small_new --encoder small --steps 40
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_upd -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 --farm-role updater > $O/trace_upd.log 2>&1
(cd $R && python scripts/chain_trace.py $O/trace_upd > $O/launches_farm_updater.txt 2>&1)
find $O -name '*.db' -delete; find $O -name '*.csv' -size +2M -delete
tail -60 $O/launches_farm_updater.txt | cut -c1-150
