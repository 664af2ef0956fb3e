mkdir -p gpurun_out/ab
for k in 0 8 16 24 32 48 64; do
  timeout 200 python bench.py --no-cpu-baseline --steps 60 --reserve-cus $k > gpurun_out/ab/cu_$k.json 2>gpurun_out/ab/err_$k.txt
  python -c "
import json;d=json.loads(open('gpurun_out/ab/cu_$k.json').read().strip().splitlines()[-1]);print('reserve $k', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/err_$k.txt
done
