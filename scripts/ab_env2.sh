#!/bin/bash
# GPU box: pipelined-only same-call A/B of one environment switch.  usage: ab_env2.sh OUT VAR "v1 v2 ..." ["bench args"]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
VAR=$2; VALS=$3; EXTRA=${4:-}
NB="--no-cpu-baseline --no-verify --steps 100 --repeats 3 $EXTRA"
for v in $VALS; do
  tag=${VAR}_${v}
  env $VAR=$v python bench.py $NB > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.split("/")[-1]: round(v["avg_us"], 1) for k, v in pk.items() if k in ("conv_init", "conv_igemm/b0_conv0", "conv_igemm/b0_conv1", "conv_igemm/b1_conv1", "conv_igemm/b2_conv1", "conv_igemm/b3_conv1")}
    print("$tag", d["value"], d["ms_per_step"], d["ms_per_step_runs"], d["roofline"]["frac"], sel)
except Exception as e:
    print("$tag FAILED", e)
PY
done
