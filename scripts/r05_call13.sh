#!/bin/bash
# Round 5, GPU call 13: anti-phase start (SERL_RS_STAGGER) re-tuned for the LDS-DMA row-slab kernels; trunk tests on the new default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call13; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_agent_gpu.py -m gpu -q -x -k "trunk or race_free or lds_dma or fused_projection or pipelined" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-250
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env $ENVV timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.replace("conv_igemm/", ""): round(v["avg_us"], 1) for k, v in pk.items() if k.startswith("conv_igemm/b0") or k.startswith("conv_igemm/b1_conv1")}
    print("$tag", d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], sel)
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-600:])
PY
}
for st in 5 0 2 3 4 6 8 5; do ENVV="SERL_RS_STAGGER=$st"; run stagger_$st; done
