#!/bin/bash
# Round 5, GPU call 19: the pipelined tests on the gather-stream default (learner classes, the DrQ agent's own prefetch, threads inserting
# during updates, the launcher), then the evidence collection on the final code
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call19; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_drq_agent_gpu.py tests/test_replay_threads_gpu.py tests/test_timed_shapes_gpu.py tests/test_bench_launcher_gpu.py tests/test_dp_two_process_gpu.py tests/test_golden_update_gpu.py tests/test_agent_gpu.py -m gpu -q -x --durations=5 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -9 $O/pytest.log | cut -c1-200
bash scripts/r05_evidence.sh > $O/evidence.log 2>&1
tail -6 $O/evidence.log | cut -c1-600
