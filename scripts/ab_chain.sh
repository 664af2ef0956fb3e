#!/bin/bash
# GPU box: same-call A/B of the update chain variants (pipelined, one rank's share of 8, serial)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-abchain}; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --steps 100 --repeats 3"
for v in "0 1" "1 0" "1 1"; do
  set -- $v; f=$1; l=$2
  for m in "" "--emulate-world 8" "--no-pipeline"; do
    tag=f${f}l${l}$(echo $m | tr -d ' -')
    SERL_CHAIN_FUSE=$f SERL_CHAIN_LN_EPI=$l python bench.py $NB $m > $O/$tag.json 2> $O/$tag.err
    python -c "
import json,sys
try:
    d=json.load(open('$O/$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['ms_per_step_runs'])
except Exception as e: print('$tag FAILED', e)
"
  done
done
