#!/bin/bash
# Round 6, GPU call 21: which stream gets the high-priority queue in the pipelined schedule (bench.py --prio; "trunk" since round 2) -- re-checked on this round's kernels,
# where the per-queue timeline shows BOTH streams ~95 % busy (the update chain's 0.69 ms of work takes 2.3 ms next to the trunk pass)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call21; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --steps 110 --repeats 3"
for rep in 1 2; do for p in trunk update none; do
  timeout 200 python bench.py $NB --prio $p > $O/${p}_$rep.json 2> $O/${p}_$rep.err
  python -c "
import json; d=json.load(open('$O/${p}_$rep.json')); print('prio $p rep $rep', d['ms_per_step'], d['ms_per_step_runs'])"
done; done
