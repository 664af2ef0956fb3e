mkdir -p gpurun_out/ab
run() { tag=$1; shift; "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/$tag.err; }
for b in 256 512 1024 4096; do
export SERL_GEMM_BLOCKS=$b
run e8_b$b timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 8
run n1_b$b timeout 200 python bench.py --no-cpu-baseline --steps 60
done
