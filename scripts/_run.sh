timeout 600 python -m pytest tests/test_agent_gpu.py -x -q -k "trunk_forward or full_size or pipelined" 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -2
timeout 200 python bench.py --no-cpu-baseline --steps 40 | tail -1 | cut -c1-200
