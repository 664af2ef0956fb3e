#!/bin/bash
# GPU box: same-call A/B of the system-scope fence of the library's / the schedule's ordering events
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_fence; mkdir -p $O; cd $R
for v in 1 0 1 0; do
  for m in "--no-pipeline" ""; do
    for np in 1 0; do
      tag=fence${v}_noprof${np}$(echo $m | tr -d ' -')
      SERL_EVENT_FENCE=$v SERL_BENCH_NOPROF=$np python bench.py --no-cpu-baseline --no-verify --steps 100 --repeats 2 $m > $O/$tag.json 2> $O/$tag.err
      python -c "
import json
try:
    d = json.load(open('$O/$tag.json')); print('$tag', d['ms_per_step'], d['ms_per_step_runs'])
except Exception as e: print('$tag FAILED', e)"
    done
  done
done
