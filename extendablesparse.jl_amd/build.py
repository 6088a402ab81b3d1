"""Build recipe of libesparse_hip.so (hipcc, gfx950 only, in-tree).

Every csrc/*.hip is a translation unit of its own (the ~50 instantiations of the bucket kernel live in
local_*.hip): they compile in parallel into csrc/build/*.o and are linked into one shared library.  A unit is
rebuilt when it or a header it included last time (hipcc -MD) is newer than its object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
SO = os.path.join(HERE, "libesparse_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off",  # value streams must match the oracle bit for bit (no FMA contraction)
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def units():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))


def _obj(unit):
    return os.path.join(OBJ, os.path.basename(unit)[:-4] + ".o")


def _extra():
    return os.environ.get("ESP_EXTRA_FLAGS", "").split()  # e.g. -DESP_LOCAL_STAMPS for the phase-stamp diagnostics


def _deps(unit):
    """Files the unit's object depends on: from the compiler's dependency file when there is one, else every source."""
    d = _obj(unit)[:-2] + ".d"
    try:
        with open(d) as f:
            txt = f.read().replace("\\\n", " ")
        files = txt.split(":", 1)[1].split()
        return [p for p in files if not p.startswith(("/opt/", "/usr/"))] or sources()
    except Exception:
        return sources() + [os.path.join(os.path.dirname(HERE), "include", "esparse_hip.h")]


def _stale(unit):
    o = _obj(unit)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    stamp = o[:-2] + ".flags"
    try:
        with open(stamp) as f:
            if f.read() != " ".join(FLAGS + _extra()):
                return True
    except Exception:
        return True
    return any((not os.path.exists(p)) or os.path.getmtime(p) > t for p in _deps(unit))


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(_stale(u) or os.path.getmtime(_obj(u)) > t for u in units())


def _compile(unit, verbose):
    o = _obj(unit)
    cmd = [HIPCC] + FLAGS + _extra() + ["-c", "-MD", "-MF", o[:-2] + ".d", "-o", o, unit]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(o[:-2] + ".flags", "w") as f:
        f.write(" ".join(FLAGS + _extra()))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    os.makedirs(OBJ, exist_ok=True)
    todo = [u for u in units() if force or _stale(u)]
    jobs = max(1, min(len(todo), int(os.environ.get("ESP_BUILD_JOBS", str(os.cpu_count() or 4)))))
    if todo:
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(lambda u: _compile(u, verbose), todo))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + [_obj(u) for u in units()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv or len(sys.argv) == 1, verbose=True)
