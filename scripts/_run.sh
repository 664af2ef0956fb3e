mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_agent_gpu.py -x -q > gpurun_out/t.log 2>&1; echo "tests rc=$?" >> gpurun_out/t.log; tail -4 gpurun_out/t.log
for b in 64 128 256 512 100000; do
  SERL_GEMM_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/ab/gb_$b.json 2>gpurun_out/ab/err_$b.txt
  python -c "
import json;d=json.loads(open('gpurun_out/ab/gb_$b.json').read().strip().splitlines()[-1]);print('budget $b', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/err_$b.txt
done
for b in 256 100000; do
  SERL_GEMM_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline --steps 60 --emulate-world 8 > gpurun_out/ab/gb8_$b.json 2>gpurun_out/ab/err8_$b.txt
  python -c "
import json;d=json.loads(open('gpurun_out/ab/gb8_$b.json').read().strip().splitlines()[-1]);print('emu8 budget $b', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/err8_$b.txt
done
SERL_GEMM_BLOCKS=256 timeout 200 python bench.py --no-cpu-baseline --steps 60 --no-pipeline > gpurun_out/ab/gbs.json 2>gpurun_out/ab/errs.txt
python -c "
import json;d=json.loads(open('gpurun_out/ab/gbs.json').read().strip().splitlines()[-1]);print('serial', d['value'], d['ms_per_step'])"
