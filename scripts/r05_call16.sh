#!/bin/bash
# Round 5, GPU call 16: the update chain's stream confined to a CU mask (SERL_UPDATE_CUS = number of CUs, spread over the XCDs) -- does
# concentrating the chain's workgroups on a part of the chip stretch the co-running trunk pass less?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call16; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env $ENVV timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.replace("conv_igemm/", ""): round(v["avg_us"], 1) for k, v in pk.items() if k in ("conv_init", "conv_igemm/b0_conv0", "conv_igemm/b0_conv1", "conv_igemm/b3_conv1", "adam_ema")}
    print("$tag", d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], sel)
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-600:])
PY
}
for n in 0 128 64 32 0 96 16; do ENVV="SERL_UPDATE_CUS=$n"; run cus_$n; done
