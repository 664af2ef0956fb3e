"""CPU: the C-ABI library builds, loads and exports every symbol include/serl_mi355.h declares."""
import ctypes
import os


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
