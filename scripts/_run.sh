mkdir -p gpurun_out/ab
run() { tag=$1; shift; "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/$tag.err; }
for rep in 1 2; do
SERL_PG_DEFER_ROWS=256 run n1_defer$rep timeout 200 python bench.py --no-cpu-baseline --steps 60
run n1_direct$rep timeout 200 python bench.py --no-cpu-baseline --steps 60
done
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo rc=$? >> gpurun_out/t.log; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" gpurun_out/t.log | tail -3
