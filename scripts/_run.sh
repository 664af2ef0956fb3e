mkdir -p gpurun_out/ab
run() { tag=$1; shift; "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/$tag.err; }
for w in 8 4 2; do
run mid${w}_512 timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world $w
SERL_CONV_MID_MIN=0 run mid${w}_off timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world $w
done
timeout 300 python -m pytest tests/test_agent_gpu.py -x -q -k "trunk_forward or full_size or dp_split" 2>&1 | tail -2
