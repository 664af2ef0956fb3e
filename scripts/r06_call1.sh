#!/bin/bash
# Round 6, GPU call 1: the full GPU suite with every BASELINE config's full-shape parity un-gated, the default bench line, and the chain's pieces (baseline of the round)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call1; rm -rf $O; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log | cut -c1-250
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; head -c 400 $O/bench.json; echo
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
for v in "farm_updater --farm-role updater" "serial --no-pipeline" "emu8 --emulate-world 8"; do
  set -- $v; tag=$1; shift
  timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', d.get('ms_per_step', d.get('diagnostic_ms_per_step')), d['ms_per_step_runs'])"
done
