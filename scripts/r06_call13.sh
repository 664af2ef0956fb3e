#!/bin/bash
# Round 6, GPU call 13: TIMING ABLATIONS of conv_dma_f16x3_kernel (the LDS-DMA ring kernel of b1_conv0, b2_*, b3_*; side libraries built with -DABL=mask, wrong results,
# never shipped): 1 no epilogues, 2 no MFMAs, 4 no LDS fragment reads, 8 no DMA pieces.  Serial schedule, HIP-event time per launch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call13; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --no-pipeline --steps 60 --repeats 2"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); pk=d['roofline']['per_kernel']
    print('$tag', d.get('ms_per_step'), {k.split('/')[-1]: round(v['avg_us']) for k, v in pk.items() if 'b1_conv0' in k or 'b2_' in k or 'b3_' in k})
except Exception as e: print('$tag FAILED', e, open('$O/$tag.err').read()[-400:])"; }
run full
for m in 1 2 4 8 3 6 14 15; do SERL_MI355_LIB=$R/serl_amd/lib/libabl_$m.so run abl_$m; done
run full_again
