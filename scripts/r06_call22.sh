#!/bin/bash
# Round 6, GPU call 22: the bf16-split timing ablation (call 3) in the PIPELINED schedule: the chain runs in what the conv workgroups leave free, so its instruction
# count may weigh more there than alone (-4 % alone).  Also SERL_GEMM=f32 (no split at all, 8 slower MFMAs) for comparison.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call22; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
for rep in 1 2; do
  timeout 200 python bench.py $NB > $O/full_$rep.json 2> $O/full_$rep.err
  SERL_MI355_LIB=$R/serl_amd/lib/libabl_split.so timeout 200 python bench.py $NB > $O/abl_$rep.json 2> $O/abl_$rep.err
  SERL_GEMM=f32 timeout 200 python bench.py $NB > $O/f32_$rep.json 2> $O/f32_$rep.err
  for t in full abl f32; do python -c "
import json; d=json.load(open('$O/${t}_$rep.json')); print('$t rep $rep', d['ms_per_step'], d['ms_per_step_runs'])"; done
done
