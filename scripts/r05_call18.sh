#!/bin/bash
# Round 5, GPU call 17: s_setprio 3 in the trunk's conv kernels (SERL_TRUNK_WPRIO=1): the trunk is the critical path, the update
# chain's waves that share a SIMD with it are not -- does the arbiter's preference shrink the co-run stretch?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call18b; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env $ENVV timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.replace("conv_igemm/", ""): round(v["avg_us"], 1) for k, v in pk.items() if k in ("conv_init", "conv_igemm/b0_conv0", "conv_igemm/b0_conv1", "conv_igemm/b1_conv0", "conv_igemm/b3_conv1", "adam_ema")}
    print("$tag", d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], sel)
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-600:])
PY
}
for v in 0 1 0 1; do ENVV="SERL_GATHER_STREAM=$v"; run gstream_$v; done
