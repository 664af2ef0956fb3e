#!/bin/bash
# Round 5, GPU call 6: the two designs for N GPUs, measured piece by piece on ONE GPU in one call.
#   trunk farm: worker role (gather + augment + full-batch trunk, no update), updater role (update chain alone, features by D2D copy)
#   batch-sharded DP: one rank's share of the step at world 2 / 4 / 8 (no collective), and at world 8 with the RCCL calls issued
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call6; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_dp_two_process_gpu.py -m gpu -q > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log | cut -c1-300
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    ms = d.get("ms_per_step", d.get("diagnostic_ms_per_step"))
    print("$tag", ms, d["ms_per_step_runs"], d["config"]["parallelism"])
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-800:])
PY
}
run one_gpu
run farm_worker --farm-role worker
run farm_worker_serial --farm-role worker --no-pipeline
run farm_updater --farm-role updater
run farm_updater_serial --farm-role updater --no-pipeline
run farm_updater_hash --farm-role updater --noise hash
run dp_emu2 --emulate-world 2
run dp_emu4 --emulate-world 4
run dp_emu8 --emulate-world 8
run dp_emu8_coll --emulate-world 8 --force-collective
run one_gpu_b
