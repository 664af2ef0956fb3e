#!/bin/bash
# GPU box: kernel trace of the pipelined bench -> gpurun_out/$1/timeline.txt (scripts/timeline_full.py: one step, both queues with start offsets)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tl}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 ${2:-} > $O/trace.log 2>&1
cd $R; python scripts/timeline_full.py $O/trace > $O/timeline.txt 2>&1; python scripts/chain_trace.py $O/trace > $O/launches.txt 2>&1; find $O -name "*.csv" -size +1M -delete; tail -3 $O/timeline.txt
