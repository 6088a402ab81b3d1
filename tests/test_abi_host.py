"""A C99 host of the C ABI (tests/abi_host.c): the shim's call sequences from a language other than Python, through
include/esparse_hip.h itself.  Without a GPU: it compiles as strict C99 and links against the library; with one: it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "extendablesparse.jl_amd")
EXE = os.path.join(ROOT, "tests", "abi_host.bin")


def build_host():
    import importlib.util
    spec = importlib.util.spec_from_file_location("esparse_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_host.c"), "-o", EXE, "-L", PKG, "-l:libesparse_hip.so", "-ldl",
           "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_abi_host_compiles_as_c99():
    exe = build_host()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_abi_host_runs():
    exe = build_host()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "abi_host: ok" in r.stdout
