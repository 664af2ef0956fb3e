timeout 600 python -m pytest tests/test_agent_gpu.py -x -q -k "negative_and_zero" 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -8
