#!/bin/bash
# Round 5, final state: the full GPU suite as the FIRST process of a fresh box (durations), the long variants (SERL_SLOW=1) on their
# own, then the evidence collection (scripts/r05_evidence.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_suite; rm -rf $O; mkdir -p $O; cd $R
date +%s > $O/t0
timeout 1000 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
date +%s > $O/t1
tail -36 $O/pytest_gpu.log | cut -c1-200
echo "suite $(( $(cat $O/t1) - $(cat $O/t0) )) s"
SERL_SLOW=1 timeout 900 python -m pytest tests -m "gpu and slow" -q --durations=10 > $O/pytest_gpu_slow.log 2>&1
echo "pytest slow rc=$?" >> $O/pytest_gpu_slow.log
tail -14 $O/pytest_gpu_slow.log | cut -c1-200
bash scripts/r05_evidence.sh > $O/evidence.log 2>&1
tail -12 $O/evidence.log | cut -c1-700
