"""CPU: the C-ABI library builds, loads and exports every symbol include/serl_mi355.h declares."""
import ctypes
import os

import pytest


def test_library_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from serl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.exported_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert L.serl_version() >= 100


def test_product_path_has_no_oracle_import():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dp, _, files in os.walk(os.path.join(root, "serl_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if "import oracle" in txt or "from oracle" in txt:
                    bad.append(f)
    assert not bad, bad


@pytest.mark.gpu
def test_plain_c_client_of_the_abi(gpu, tmp_path):
    """tests/c_abi_smoke.c is C99 and includes nothing but the HIP runtime API and include/serl_mi355.h: it builds a
    state-only SAC learner (replay buffer in HBM -> fused gather -> update_high_utd) through the C ABI alone."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.join(root, "serl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", os.path.join(root, "tests", "c_abi_smoke.c"), "-I" + os.path.join(root, "include"),
                           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + libdir, "-lserl_mi355", "-L/opt/rocm/lib",
                           "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "C ABI OK" in out.stdout, (out.stdout, out.stderr)
