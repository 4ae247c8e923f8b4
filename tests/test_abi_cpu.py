"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/dct.h declares, and the ctypes binding table covers exactly that set."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dct.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dct_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dct_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) > 40
    for n in names:
        assert hasattr(lib, n), f"libdct_hip.so does not export {n}"


def test_binding_table_matches_header():
    from dct_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_status_strings_and_version():
    from dct_amd import _lib
    lib = _lib.load()
    assert lib.dct_version() >= 100
    assert lib.dct_status_string(0) == b"ok"
    assert b"workspace" in lib.dct_status_string(-4)


def test_product_path_rejects_cpu_tensors():
    import pytest
    import torch
    from dct_amd import _lib
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.view(torch.zeros(1, 2, 2, 8))
    with pytest.raises(RuntimeError):
        _lib.check(-2, "x")
