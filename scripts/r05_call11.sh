#!/bin/bash
# Round 5: the full GPU suite as the FIRST process of a fresh box (durations), then the evidence collection
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_suite; rm -rf $O; mkdir -p $O; cd $R
date +%s > $O/t0
timeout 1000 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
date +%s > $O/t1
tail -40 $O/pytest_gpu.log | cut -c1-200
echo "suite $(( $(cat $O/t1) - $(cat $O/t0) )) s"
bash scripts/r05_evidence.sh > $O/evidence.log 2>&1
tail -12 $O/evidence.log | cut -c1-700
