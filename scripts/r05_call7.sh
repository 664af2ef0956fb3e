#!/bin/bash
# Round 5, GPU call 7: K-split budget of the update chain when NOTHING co-runs (the trunk farm's updater), and a HIP-API + kernel trace of the pass boundary
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call7; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env $ENVV timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    print("$tag", d.get("ms_per_step", d.get("diagnostic_ms_per_step")), d["ms_per_step_runs"])
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-800:])
PY
}
ENVV="X=0"; run upd_default --farm-role updater
ENVV="SERL_SPLIT_BUDGET=1024"; run upd_b1024 --farm-role updater
ENVV="SERL_SPLIT_BUDGET=2048"; run upd_b2048 --farm-role updater
ENVV="SERL_SPLIT_BUDGET=4096"; run upd_b4096 --farm-role updater
ENVV="SERL_SPLIT_BUDGET=2048 SERL_ENC_SPLIT_BUDGET=2048"; run upd_b2048_enc --farm-role updater
ENVV="X=0"; run upd_default_b --farm-role updater
bash scripts/trace_host.sh r05_call7/th > $O/trace_host.log 2>&1; tail -70 $O/th/host_wait.txt 2>/dev/null | head -90
