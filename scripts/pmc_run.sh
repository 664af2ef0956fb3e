#!/bin/bash
# usage (on the GPU box): scripts/pmc_run.sh <tag> <counters...>   -- one PMC pass of a short bench run
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -o p -- \
  python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pipeline --fill 1200 --steps 6 --warmup 2 ${BENCH_ARGS} > $out.log 2>&1 || tail -5 $out.log
ls $out | head
