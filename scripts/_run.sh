timeout 300 python bench.py --workload sac_state --steps 200 2>/dev/null | tail -1 | cut -c1-300
