#!/bin/bash
# GPU box: per-kernel rocprofv3 stats of the serial schedule, fused and one-launch-per-operation update chain (same call)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-profchain}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PB="--no-cpu-baseline --no-verify --fill 3000 --steps 30 --warmup 5 --repeats 1 --no-pipeline ${2:-}"
for f in 1 0; do
  SERL_CHAIN_FUSE=$f timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats_f$f -o s -- python $R/bench.py $PB > $O/stats_f$f.log 2>&1
  python $R/scripts/rocprof_summary.py $(find $O/stats_f$f -name '*results.db' | head -1) $O/kernel_stats_f$f.csv
done
find $O -name '*.db' -delete; find $O -name '*.csv' -size +2M -delete
