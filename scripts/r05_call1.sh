#!/bin/bash
# Round 5, first GPU call: (1) the two new trunk tests (fused projection, depth-first chunks) + the trunk parity / race tests,
# (2) same-call A/B of SERL_PROJ_FUSE and SERL_TRUNK_CHUNK against the default (pipelined and serial), (3) the suite's durations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call1; rm -rf $O; mkdir -p $O; cd $R
date +%s > $O/t0
timeout 400 python -m pytest tests/test_agent_gpu.py -m gpu -x -q --durations=15 -k "fused_projection or depth_first or trunk_forward or race_free_at_full" > $O/pytest_new.log 2>&1
echo "rc=$?" >> $O/pytest_new.log; tail -25 $O/pytest_new.log
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {  # tag, env..., -- bench args
  tag=$1; shift
  env "$@" timeout 200 python bench.py $NB $EXTRA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.replace("conv_igemm/", ""): round(v.get("pass_us", v["avg_us"]), 1) for k, v in pk.items() if k.startswith("conv_i")}
    print("$tag", d.get("value"), d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], d["roofline"].get("frac_by_stage"), sel)
except Exception as e:
    print("$tag FAILED", e)
PY
}
EXTRA=""
run base_a X=0
run projfuse SERL_PROJ_FUSE=1
run chunk256 SERL_TRUNK_CHUNK=256
run chunk512 SERL_TRUNK_CHUNK=512
run chunk128 SERL_TRUNK_CHUNK=128
run chunk256_pf SERL_TRUNK_CHUNK=256 SERL_PROJ_FUSE=1
run base_b X=0
EXTRA="--no-pipeline"
run serial_base X=0
run serial_chunk256 SERL_TRUNK_CHUNK=256
run serial_projfuse SERL_PROJ_FUSE=1
date +%s > $O/t1
echo "elapsed $(( $(cat $O/t1) - $(cat $O/t0) )) s"
