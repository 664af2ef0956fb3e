#!/bin/bash
# Round 6, GPU call 4: the full GPU suite + the default bench line after the hygiene refactor (trunk_f16x3 split by kernel family, dead kernels / switches removed)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call4; rm -rf $O; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ) > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -22 $O/pytest.log | cut -c1-250
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
