#!/bin/bash
# Round 5, GPU call 5: learner / DP tests after the threefry key schedule moved into DataParallelLearner; A/B of --noise threefry | hash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call5; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_golden_update_gpu.py tests/test_dp_two_process_gpu.py tests/test_bench_launcher_gpu.py tests/test_variants_gpu.py tests/test_jaxrng.py -m gpu -q --durations=8 -k "threefry or two_ranks or launcher or fp32_gemm or device_draws" > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log | cut -c1-300
timeout 300 python -m pytest tests/test_agent_gpu.py -m gpu -q -k "pipelined or DataParallel or schedule" >> $O/pytest.log 2>&1; tail -4 $O/pytest.log
NB="--no-cpu-baseline --steps 110 --repeats 3"
run() {
  tag=$1; shift
  timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    print("$tag", d.get("value"), d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], (d.get("verify") or {}).get("worst_rel_diff"), d["last_info"])
except Exception as e:
    print("$tag FAILED", e, open("$O/$tag.err").read()[-600:])
PY
}
run tf_a --noise threefry
run hash_a --noise hash
run tf_b --noise threefry
run hash_b --noise hash
run serial_tf --noise threefry --no-pipeline
run serial_hash --noise hash --no-pipeline
run emu8_tf --noise threefry --emulate-world 8
run emu8_hash --noise hash --emulate-world 8
