#!/bin/bash
# GPU box: timing-only ablations of conv_init (serial schedule, per-kernel HIP-event times)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_cinit; mkdir -p $O; cd $R
for v in ${1:-0 1 2 4 8 6 7 15}; do
  SERL_CINIT_ABLATE=$v python bench.py --no-cpu-baseline --no-verify --no-pipeline --steps 60 --repeats 1 > $O/a$v.json 2> $O/a$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/a$v.json")); pk = d["roofline"]["per_kernel"]
    print("ablate=$v", d["ms_per_step"], {k: round(x["avg_us"], 1) for k, x in pk.items() if k in ("conv_init", "conv_igemm/b0_conv0")})
except Exception as e:
    print("ablate=$v FAILED", e)
PY
done
