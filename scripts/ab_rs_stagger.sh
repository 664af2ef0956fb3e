#!/bin/bash
# A/B of the row-slab kernel's anti-phase start (SERL_RS_STAGGER = number of s_sleep(127) of every CU's second workgroup):
# alternating runs in one call; each line: variant, median ms per step, the runs, per-kernel averages of the row-slab launches.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_rs_stagger; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --fill 3000 --steps 110 --repeats 2 --warmup 10"
run() {
  SERL_RS_STAGGER=$1 timeout 40 python bench.py $NB $2 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); pk = d['roofline'].get('per_kernel', {})
print('stagger=$1 $2', d['ms_per_step'], d['ms_per_step_runs'], {k.split('/')[-1]: round(v['avg_us'], 1) for k, v in pk.items() if k.split('/')[-1] in ('b0_conv0', 'b0_conv1', 'b1_conv1')})" >> $O/ab.txt
}
for s in ${STAGGERS:-0 5 3 0 8 5}; do run $s; done
run 0 --no-pipeline; run 5 --no-pipeline
cat $O/ab.txt
