#!/bin/bash
# Round 5, GPU call 3: the JAX-stream tests (device draws; HIP agent from the seed only vs the threefry goldens), the API-level DrQ tests,
# A/B of zero-copy parameter staging (does the H2D copy command cause the pass-boundary gap?) with a per-queue timeline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call3; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_jaxrng.py tests/test_golden_update_gpu.py tests/test_drq_agent_gpu.py tests/test_sac_state_gpu.py -m gpu -q --durations=8 > $O/pytest.log 2>&1
echo "rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py $NB $EXTRA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    print("$tag", d.get("value"), d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"])
except Exception as e:
    print("$tag FAILED", e)
PY
}
EXTRA=""
run base_a X=0
run zc_a SERL_STAGE_ZEROCOPY=1
run base_b X=0
run zc_b SERL_STAGE_ZEROCOPY=1
EXTRA="--no-pipeline"
run serial_base X=0
run serial_zc SERL_STAGE_ZEROCOPY=1
(cd /tmp && export TMPDIR=/tmp && SERL_STAGE_ZEROCOPY=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 > $O/trace.log 2>&1)
python scripts/timeline_streams.py $O/trace > $O/timeline_streams_zc.txt 2>&1; head -12 $O/timeline_streams_zc.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
