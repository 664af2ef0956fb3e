mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo "tests rc=$?" >> gpurun_out/t.log; tail -5 gpurun_out/t.log
run() { tag=$1; shift; "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/$tag.err; }
run n1 timeout 200 python bench.py --no-cpu-baseline --steps 60
run e8 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 8
run e2 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 2
run e4 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 4
