#!/bin/bash
# Round 6, GPU call 7: TIMING ABLATIONS of the row-slab kernels' fused epilogue (rowtile_epilogue_t; side libraries, wrong results, never shipped):
# 16 no arrive-and-wait, 32 no statistics atomics, 64 no residual loads, 128 no final stores (48 = 16 + 32, 240 = all four).  Serial schedule.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call7; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --no-pipeline --steps 60 --repeats 2"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); pk=d['roofline']['per_kernel']
    print('$tag', d.get('ms_per_step'), {k.split('/')[-1]: round(v['avg_us']) for k, v in pk.items() if 'b0_conv' in k or 'b1_conv' in k})
except Exception as e: print('$tag FAILED', e, open('$O/$tag.err').read()[-400:])"; }
run full
for m in 16 32 64 128 48 240; do SERL_MI355_LIB=$R/serl_amd/lib/libabl_$m.so run abl_$m; done
run full_again
