#!/bin/bash
# GPU box: kernel trace of `bench.py $2` -> gpurun_out/$1/trace.txt (scripts/chain_trace.py: one step, both queues, launch order)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-trace}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 ${2:-} > $O/trace.log 2>&1
cd $R; python scripts/chain_trace.py $O/trace > $O/trace.txt; find $O -name "*.csv" -size +1M -delete
