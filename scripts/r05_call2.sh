#!/bin/bash
# Round 5, GPU call 2: A/B of the insert-event wait in front of every gather (old: always; new: only while the insert is in flight),
# per-queue timeline of the new default (pass-boundary gap), then the full GPU suite with durations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_call2; rm -rf $O; mkdir -p $O; cd $R
date +%s > $O/t0
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() {
  tag=$1; shift
  env "$@" timeout 200 python bench.py $NB $EXTRA > $O/$tag.json 2> $O/$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/$tag.json"))
    pk = d["roofline"]["per_kernel"]
    sel = {k.replace("conv_igemm/", ""): round(v.get("pass_us", v["avg_us"]), 1) for k, v in pk.items() if k.startswith("conv_i")}
    print("$tag", d.get("value"), d.get("ms_per_step"), d["ms_per_step_runs"], d["roofline"]["frac"], sel)
except Exception as e:
    print("$tag FAILED", e)
PY
}
EXTRA=""
run new_a X=0
run oldwait_a SERL_RB_OLD_WAIT=1
run new_b X=0
run oldwait_b SERL_RB_OLD_WAIT=1
EXTRA="--no-pipeline"
run serial_new X=0
run serial_oldwait SERL_RB_OLD_WAIT=1
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 > $O/trace.log 2>&1)
python scripts/timeline_streams.py $O/trace > $O/timeline_streams.txt 2>&1; head -40 $O/timeline_streams.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
date +%s > $O/t1
timeout 900 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
date +%s > $O/t2
tail -55 $O/pytest_gpu.log
echo "bench part $(( $(cat $O/t1) - $(cat $O/t0) )) s, suite $(( $(cat $O/t2) - $(cat $O/t1) )) s"
