R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/tl8
cd /tmp && export TMPDIR=/tmp
SERL_BENCH_NOPROF=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl8 -o t -- python $R/bench.py --no-cpu-baseline --fill 3000 --steps 12 --warmup 4 --emulate-world 8 > $R/gpurun_out/tl8.log 2>&1
cd $R; python scripts/timeline_streams.py gpurun_out/tl8 > gpurun_out/tl8.txt 2>&1
find gpurun_out/tl8 -name "*.csv" -size +20M -delete
