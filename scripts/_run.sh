mkdir -p gpurun_out/ab
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/ab/{sys.argv[1]}.json").read().strip().splitlines()[-1])
pk=d['roofline']['per_kernel']
print(sys.argv[1], d['value'], d['ms_per_step'], ' '.join(f"{k}={v['avg_us']:.0f}" for k,v in pk.items() if not k.startswith('conv_igemm')))
PY
}
P=$GRAFT_REPO_ROOT/serl_amd/lib/libserl_prev.so
timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/ab/pool_new.json 2>gpurun_out/ab/err; show pool_new
SERL_MI355_LIB=$P timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/ab/pool_prev.json 2>gpurun_out/ab/err; show pool_prev
timeout 200 python bench.py --no-cpu-baseline --steps 60 --no-pipeline > gpurun_out/ab/pool_new_s.json 2>gpurun_out/ab/err; show pool_new_s
SERL_MI355_LIB=$P timeout 200 python bench.py --no-cpu-baseline --steps 60 --no-pipeline > gpurun_out/ab/pool_prev_s.json 2>gpurun_out/ab/err; show pool_prev_s
timeout 300 python -m pytest tests/test_agent_gpu.py -x -q -k "trunk_forward" 2>&1 | tail -2
