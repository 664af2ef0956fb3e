#!/bin/bash
# Round 5, GPU call 12: the row-slab kernel with LDS-DMA staging (SERL_SLAB_DMA=1): parity, then same-call A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call12; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_agent_gpu.py -m gpu -q -x -k "lds_dma_staging" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log | cut -c1-250
NB="--no-cpu-baseline --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env $ENVV timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.replace("conv_igemm/", ""): round(v["avg_us"], 1) for k, v in pk.items() if k.startswith("conv_i")}
    print("$tag", d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], (d.get("verify") or {}).get("worst_rel_diff"), sel)
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-600:])
PY
}
ENVV="SERL_SLAB_DMA=0"; run regs_a
ENVV="SERL_SLAB_DMA=1"; run dma_a
ENVV="SERL_SLAB_DMA=1s"; run dmas_a
ENVV="SERL_SLAB_DMA=1w"; run dmaw_a
ENVV="SERL_SLAB_DMA=1sw"; run dmasw_a
ENVV="SERL_SLAB_DMA=0"; run regs_b
ENVV="SERL_SLAB_DMA=1"; run dma_b
ENVV="SERL_SLAB_DMA=1s"; run dmas_b
ENVV="SERL_SLAB_DMA=1w"; run dmaw_b
ENVV="SERL_SLAB_DMA=1sw"; run dmasw_b
ENVV="SERL_SLAB_DMA=0"; run serial_regs --no-pipeline
ENVV="SERL_SLAB_DMA=1"; run serial_dma --no-pipeline
ENVV="SERL_SLAB_DMA=1w"; run serial_dmaw --no-pipeline
ENVV="SERL_SLAB_DMA=1sw"; run serial_dmasw --no-pipeline
