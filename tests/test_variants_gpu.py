"""GPU: the one arithmetic variant of the update chain that is still selectable at run time -- SERL_GEMM=f32, the exact
fp32-MFMA GEMM (v_mfma_f32_32x32x2_f32) instead of the default bf16x3 GEMM -- passes the same bench-shape parity test (the
switch is read once per process, so the test body runs in a child process).  Everything else that lost a measurement was
deleted in round 3."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exact_fp32_gemm_passes_the_bench_shape_parity_test(gpu):
    env = dict(os.environ, SERL_GEMM="f32")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_bench_shape_gpu.py", "-q", "-x", "-m", "gpu", "-k",
                        "update_high_utd_at_bench_shape"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "1 passed" in r.stdout, r.stdout[-500:]
