#!/bin/bash
# Round 6, GPU call 20: K-split budget of the update chain's GEMMs while it co-runs with the trunk pass (pipelined schedule; 256 since round 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call20; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --steps 110 --repeats 3"
for rep in 1 2; do for b in 256 64 128 192 384 512; do
  SERL_TMP_BUDGET=$b timeout 200 python bench.py $NB > $O/b${b}_$rep.json 2> $O/b${b}_$rep.err
  python -c "
import json; d=json.load(open('$O/b${b}_$rep.json')); print('budget $b rep $rep', d['ms_per_step'], d['ms_per_step_runs'])"
done; done
