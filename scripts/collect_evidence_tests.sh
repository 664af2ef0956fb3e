#!/bin/bash
# Third short call: one timing-only diagnostic pair (does the event record at the end of a trunk pass cost the idle time in front
# of the next pass?), then the GPU tests that scripts/collect_evidence_min.sh did not reach.  Outputs: gpurun_out/evidence_tests/.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/evidence_tests; rm -rf $O; mkdir -p $O
cd $R
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
for i in 1 2; do
  timeout 60 python bench.py $NB 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['ms_per_step'], d['ms_per_step_runs'])" >> $O/diag_noproduced.txt
  SERL_BENCH_DIAG=noproduced timeout 60 python bench.py $NB 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noproduced', d['ms_per_step'], d['ms_per_step_runs'])" >> $O/diag_noproduced.txt
done
cat $O/diag_noproduced.txt
timeout ${SUITE_TIMEOUT:-235} python -m pytest "tests/test_bench_shape_gpu.py::test_two_buffer_update_at_bench_shape" "tests/test_bench_shape_gpu.py::test_fwbw_batch_512_update" tests/test_sac_state_gpu.py tests/test_drq_agent_gpu.py tests/test_replay_gpu.py tests/test_replay_threads_gpu.py tests/test_classifier_gpu.py tests/test_abi.py tests/test_dp_two_process_gpu.py tests/test_bench_launcher_gpu.py tests/test_variants_gpu.py tests/test_chain_fusion_gpu.py -m gpu -x -v --durations=15 > $O/pytest_gpu_rest.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_rest.log
tail -30 $O/pytest_gpu_rest.log
