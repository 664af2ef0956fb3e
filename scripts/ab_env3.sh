#!/bin/bash
# GPU box: same-call A/B of one environment switch, serial schedule only, chosen per-kernel times.  usage: ab_env3.sh OUT VAR "v1 v2" "kernel,kernel,..."
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
VAR=$2; VALS=$3; KEYS=$4
for v in $VALS; do
  for m in "--no-pipeline" ""; do
    tag=${VAR}_${v}$(echo $m | tr -d ' -')
    env $VAR=$v python bench.py --no-cpu-baseline --no-verify --steps 80 --repeats 2 $m > $O/$tag.json 2> $O/$tag.err
    python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json")); pk = d["roofline"]["per_kernel"]
    print("$tag", d["ms_per_step"], d["ms_per_step_runs"], {k: round(v["avg_us"], 1) for k, v in pk.items() if any(x in k for x in "$KEYS".split(","))})
except Exception as e:
    print("$tag FAILED", e)
PY
  done
done
