#!/bin/bash
# Round 5, GPU call 14: row-major fused epilogue of the row-slab kernels (SERL_EPI_T=1) -- parity, then same-call A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call14; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_agent_gpu.py -m gpu -q -x -s -k "row_major or lds_dma" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "rel err|passed|failed|rc=|Error|assert" $O/pytest.log | cut -c1-250 | tail -12
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
for v in 0 15 0 15 1 8 4; do ENVV="SERL_EPI_T=$v"; run epi_t_$v; done
ENVV="SERL_EPI_T=0"; run serial_epi_0 --no-pipeline
ENVV="SERL_EPI_T=15"; run serial_epi_15 --no-pipeline
