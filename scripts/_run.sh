timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1; echo rc=$? >> gpurun_out/t.log; grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" gpurun_out/t.log | tail -6
