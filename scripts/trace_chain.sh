#!/bin/bash
# GPU box: per-launch view of one serial step (scripts/chain_trace.py) -> gpurun_out/$1/chain_trace_serial.txt ; extra bench args in $2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 --no-pipeline ${2:-} > $O/trace.log 2>&1
cd $R; python scripts/chain_trace.py $O/trace > $O/chain_trace_serial.txt; find $O -name "*.csv" -size +1M -delete
