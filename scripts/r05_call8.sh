#!/bin/bash
# Round 5, GPU call 8: jax.random draws INSIDE the consuming kernels (serl_noise keys) -- parity tests, then the cost of the three forms
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call8; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_jaxrng.py tests/test_golden_update_gpu.py tests/test_drq_agent_gpu.py tests/test_chain_fusion_gpu.py tests/test_dp_two_process_gpu.py tests/test_sac_state_gpu.py -m gpu -q --durations=6 > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log | cut -c1-300
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env $ENVV timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    print("$tag", d.get("ms_per_step", d.get("diagnostic_ms_per_step")), d["ms_per_step_runs"], d["last_info"].get("critic_loss"))
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-800:])
PY
}
ENVV="X=0"
run keys_a
run hash_a --noise hash
run keys_b
run hash_b --noise hash
run serial_keys --no-pipeline
run serial_hash --no-pipeline --noise hash
run emu8_keys --emulate-world 8
run emu8_hash --emulate-world 8 --noise hash
run upd_keys --farm-role updater
run upd_hash --farm-role updater --noise hash
