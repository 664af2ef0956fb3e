mkdir -p gpurun_out/ab
run() { tag=$1; shift; "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'])" || tail -3 gpurun_out/ab/$tag.err; }
run n1 timeout 200 python bench.py --no-cpu-baseline
run e8 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 8
SERL_BENCH_NOPROF=1 run e8_noprof timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 8
run e4 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 4
run e2 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 2
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
