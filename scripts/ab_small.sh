#!/bin/bash
# GPU box: SmallEncoder golden tests + same-call A/B of the layer >= 1 weight-gradient kernel (SERL_SMALL_WGRAD_MFMA = last layer on the fp32 MFMA kernel)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_small; mkdir -p $O; cd $R
for v in ${1:-0 1 2 3}; do
  SERL_SMALL_WGRAD_MFMA=$v timeout 600 python -m pytest tests/test_small_encoder_gpu.py -x -q 2>&1 | tail -2
  SERL_SMALL_WGRAD_MFMA=$v python bench.py --encoder small --no-cpu-baseline --no-verify --steps 50 --repeats 3 > $O/m$v.json 2> $O/m$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/m$v.json")); pk = d["roofline"].get("per_kernel", {})
    print("mfma<=$v", d["value"], d["ms_per_step"], d["ms_per_step_runs"], {k: round(x["avg_us"], 1) for k, x in pk.items() if "small" in k or "wgrad" in k})
except Exception as e:
    print("mfma<=$v FAILED", e)
PY
done
