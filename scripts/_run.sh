timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo rc=$? >> gpurun_out/t.log; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" gpurun_out/t.log | tail -4
SERL_POOL_FUSE=0 timeout 300 python -m pytest tests/test_agent_gpu.py -x -q -k "trunk_forward" 2>&1 | tail -2
