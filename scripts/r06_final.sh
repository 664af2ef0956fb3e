#!/bin/bash
# Round 6, last GPU call: the full GPU suite on the final code (first process of a fresh box), then scripts/r06_evidence.sh on the same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_final; rm -rf $O; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 ) > $O/pytest_gpu.log 2>&1
echo "rc=$?" >> $O/pytest_gpu.log; tail -24 $O/pytest_gpu.log | cut -c1-200
bash scripts/r06_evidence.sh $1
