#!/bin/bash
# GPU box: HIP runtime API trace + kernel trace of the pipelined bench: what does the host wait for at a pass boundary?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-th}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 10 --warmup 5 --repeats 1 ${2:-} > $O/trace.log 2>&1
cd $R; ls -la $O/trace | head; python scripts/host_wait.py $O/trace > $O/host_wait.txt 2>&1; tail -60 $O/host_wait.txt; find $O -name "*.csv" -size +1M -delete
