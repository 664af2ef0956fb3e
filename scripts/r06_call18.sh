#!/bin/bash
# Round 6, GPU call 18: EXPERIMENT -- unscaled lo planes (kLoScale = 1: lo = fp16(x - hi), subnormal for small values): do the fp16 MFMAs keep subnormal inputs, and
# does the trunk stay within 5e-6 of fp64?  (The precondition of a single-accumulator split: hi*hi + hi*lo + lo*hi in ONE fp32 accumulator.)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call18; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_agent_gpu.py tests/test_drq_agent_gpu.py -m gpu -q -s -k "trunk_forward or pretrained_like or imagenet_like or row_slab or negative_and_zero" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; grep -E "rel err|passed|failed|rc=|assert|Error" $O/pytest.log | cut -c1-200 | tail -40
