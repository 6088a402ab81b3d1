"""CPU: the C-ABI library loads, exports every symbol include/esparse_hip.h declares, and refuses
to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "esparse_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(esp_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(esp):
    so = esp.library_path()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libesparse_hip.so does not export %s" % n
    # and the Python binding table covers exactly the header
    assert sorted(esp._lib.SIGNATURES) == names


def test_version_string(esp):
    assert b"gfx950" in esp._lib.load().esp_version()


def test_no_cpu_fallback_without_gpu(esp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(esp.NoDeviceError):
        esp.ExtendableSparseMatrix(10, 10)
    with pytest.raises(esp.NoDeviceError):
        esp.SparseMatrixHIPCOO(10, 10)


def test_product_never_imports_oracle(esp):
    """The package must not reference oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.dirname(esp.library_path())
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".jl")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "esparse_oracle.h" not in src, f


def test_host_csc_findindex(esp):
    """findindex (sparsematrixcsc.jl:7-23) of the host container used by the Generic wrappers."""
    import numpy as np
    c = esp.SparseMatrixCSC(4, 3, np.array([1, 3, 3, 5]), np.array([2, 4, 1, 3]), np.array([1., 2., 3., 4.]))
    assert c.findindex(2, 1) == 1 and c.findindex(4, 1) == 2 and c.findindex(3, 1) == 0
    assert c.findindex(1, 2) == 0 and c.findindex(1, 3) == 3 and c.findindex(3, 3) == 4
    assert c[4, 1] == 2.0 and c[2, 2] == 0.0 and c.nnz() == 4
    with pytest.raises(IndexError):
        c.findindex(5, 1)
    with pytest.raises(IndexError):
        c.findindex(1, 0)
