#!/bin/bash
# Round 6, GPU call 5: counted DMA waits in conv3x3_slabdma_f16x3_kernel (the next channel group's slab pieces stay in flight across sub-chunks):
# trunk parity / race tests, then same-call A/B (SERL_SD_WAIT=0 = the flat vmcnt(0) of round 5)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call5; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_agent_gpu.py -m gpu -q -x -k "trunk or race_free or row_slab or fused_projection or pipelined or full_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log | cut -c1-250
NB="--no-cpu-baseline --steps 110 --repeats 3"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); pk=d['roofline']['per_kernel']
    print('$tag', d.get('ms_per_step'), d['ms_per_step_runs'], {k.split('/')[-1]: round(v['avg_us']) for k, v in pk.items() if 'b0_conv' in k or 'b1_conv' in k}, d.get('verify', {}).get('worst_rel_diff'))
except Exception as e: print('$tag FAILED', e, open('$O/$tag.err').read()[-600:])"; }
for rep in 1 2; do
  SERL_SD_WAIT=0 run flat_pipe_$rep
  run counted_pipe_$rep
  SERL_SD_WAIT=0 run flat_serial_$rep --no-pipeline
  run counted_serial_$rep --no-pipeline
done
