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


def _child(test_file, k, n):
    env = dict(os.environ, SERL_GEMM="f32")
    r = subprocess.run([sys.executable, "-m", "pytest", test_file, "-q", "-x", "-m", "gpu", "-k", k],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{n} passed" in r.stdout, r.stdout[-500:]


def test_exact_fp32_gemm_passes_the_update_parity_tests(gpu):
    """critic + actor / temperature updates at B = 40 and 16, 64x64 (a few seconds; the bench-shape form below is the long one)"""
    _child("tests/test_agent_gpu.py", "test_update_high_utd_matches_oracle and f16x3", 2)


@pytest.mark.slow
def test_exact_fp32_gemm_passes_the_bench_shape_parity_test(gpu):
    _child("tests/test_bench_shape_gpu.py", "update_high_utd_at_bench_shape", 1)
