mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_agent_gpu.py tests/test_sac_state_gpu.py -x -q > gpurun_out/t.log 2>&1; echo rc=$? >> gpurun_out/t.log; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" gpurun_out/t.log | tail -3
run() { tag=$1; shift; "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/ab/$tag.json').read().strip().splitlines()[-1]);print('$tag', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/ab/$tag.err; }
P=$GRAFT_REPO_ROOT/serl_amd/lib/libserl_prev.so
for rep in 1 2; do
run e8_new$rep timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 8
SERL_MI355_LIB=$P run e8_prev$rep timeout 200 python bench.py --no-cpu-baseline --steps 100 --emulate-world 8
done
