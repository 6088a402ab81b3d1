"""Build recipe of libesparse_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libesparse_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",  # value streams must match the oracle bit for bit (no FMA contraction)
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith((".hip", ".hpp")))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    hdr = os.path.join(os.path.dirname(HERE), "include", "esparse_hip.h")
    return any(os.path.getmtime(s) > t for s in sources() + [hdr])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    extra = os.environ.get("ESP_EXTRA_FLAGS", "").split()  # e.g. -DESP_LOCAL_STAMPS for the phase-stamp diagnostics
    cmd = [HIPCC] + FLAGS + extra + ["-o", SO, os.path.join(CSRC, "esparse_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force=True, verbose=True)
