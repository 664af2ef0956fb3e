#!/bin/bash
# Round 5, last GPU call: the full GPU suite on the final code (first process of a fresh box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_suite_final; rm -rf $O; mkdir -p $O; cd $R
timeout 320 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -20 $O/pytest_gpu.log | cut -c1-160
