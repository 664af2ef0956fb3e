#!/bin/bash
# Round 6, GPU call 14: TIMING ABLATIONS of the ring kernel MAIN epilogue (dma_tile_epilogue): 16 no statistics atomics, 32 no raw stores (unfused), 64 no residual loads (fused), 128 no winv load, 256 no arrive-and-wait (fused)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call14; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --no-pipeline --steps 60 --repeats 2"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); pk=d['roofline']['per_kernel']
    print('$tag', d.get('ms_per_step'), {k.split('/')[-1]: round(v['avg_us']) for k, v in pk.items() if 'b1_conv0' in k or 'b2_' in k or 'b3_' in k})
except Exception as e: print('$tag FAILED', e, open('$O/$tag.err').read()[-400:])"; }
run full
for m in 16 32 64 128 256 48 496; do SERL_MI355_LIB=$R/serl_amd/lib/libabl_$m.so run abl_$m; done
run full_again
