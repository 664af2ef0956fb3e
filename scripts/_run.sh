timeout 300 python scripts/bench_sac_state.py 200 > gpurun_out/sac_state.json 2> gpurun_out/sac_state.err; tail -1 gpurun_out/sac_state.json; tail -3 gpurun_out/sac_state.err
