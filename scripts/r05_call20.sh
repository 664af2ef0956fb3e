#!/bin/bash
# Round 5, GPU call 20: the trunk farm on the gather-stream schedule (two-process bit-identity test, the two roles alone)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call20; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_dp_two_process_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log | cut -c1-200
NB="--no-cpu-baseline --steps 110 --repeats 3"
timeout 100 python bench.py $NB --farm-role worker > $O/bench_farm_worker.json 2> /dev/null
timeout 100 python bench.py $NB --farm-role updater > $O/bench_farm_updater.json 2> /dev/null
SERL_GATHER_STREAM=0 timeout 100 python bench.py $NB --farm-role worker > $O/bench_farm_worker_g0.json 2> /dev/null
SERL_GATHER_STREAM=0 timeout 100 python bench.py $NB > $O/bench_gather_on_trunk_stream.json 2> /dev/null
timeout 100 python bench.py $NB > $O/bench_default.json 2> /dev/null
python - <<PY
import json
for n in ("bench_farm_worker", "bench_farm_worker_g0", "bench_farm_updater", "bench_gather_on_trunk_stream", "bench_default"):
    d = json.loads(open("$O/" + n + ".json").read().strip().splitlines()[-1])
    print(n, d.get("ms_per_step", d.get("diagnostic_ms_per_step")), d.get("ms_per_step_runs"))
PY
