timeout 900 python -m pytest tests/test_sac_state_gpu.py -x -q > gpurun_out/t.log 2>&1; echo rc=$? >> gpurun_out/t.log; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" gpurun_out/t.log | tail -30
