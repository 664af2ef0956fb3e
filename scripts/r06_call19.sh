#!/bin/bash
# Round 6, GPU call 19: UNSCALED lo planes + ONE accumulator in the LDS-DMA ring kernel, and THREE ring positions / three workgroups per CU for its non-projection
# form (b2_conv1, b3_conv1; with SERL_PROJ_FUSE=0 also the conv0s): trunk parity + race tests, then same-call A/B against the previous commit's library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call19; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_agent_gpu.py tests/test_drq_agent_gpu.py -m gpu -q -x -k "trunk or race_free or row_slab or fused_projection or pipelined or full_size or k_split or imagenet_like" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log | cut -c1-250
NB="--no-cpu-baseline --steps 110 --repeats 3"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); pk=d['roofline']['per_kernel']
    print('$tag', d.get('ms_per_step'), d['ms_per_step_runs'], {k.split('/')[-1]: round(v['avg_us']) for k, v in pk.items() if 'conv' in k or 'proj' in k}, d.get('verify', {}).get('worst_rel_diff'))
except Exception as e: print('$tag FAILED', e, open('$O/$tag.err').read()[-600:])"; }
for rep in 1 2; do
  SERL_MI355_LIB=$R/serl_amd/lib/libserl_mi355_head.so run head_pipe_$rep
  SERL_RING3=0 run acc1_pipe_$rep
  run ring3_pipe_$rep
  SERL_MI355_LIB=$R/serl_amd/lib/libserl_mi355_head.so run head_serial_$rep --no-pipeline
  SERL_RING3=0 run acc1_serial_$rep --no-pipeline
  run ring3_serial_$rep --no-pipeline
done
SERL_PROJ_FUSE=0 SERL_RING3=0 run unfproj_ring4_serial --no-pipeline
SERL_PROJ_FUSE=0 run unfproj_ring3_serial --no-pipeline
