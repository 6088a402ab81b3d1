"""The exchange policy of esp_group_flush (csrc/group_policy.hpp -- the code the library itself runs) on the CPU:
tests/group_policy_test.cpp drives it with 1 / 2 / 3 / 8 ranks as threads over a host model of a shard and an in-process
transport, built with -fsanitize=address,undefined."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "group_policy_test.bin")


def _build():
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", "-pthread", os.path.join(ROOT, "tests", "group_policy_test.cpp"), "-o", EXE]
    subprocess.check_call(cmd)


def test_group_policy_threads_as_ranks_under_sanitizers():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "group_policy_test: ok" in r.stdout


@pytest.mark.parametrize("procs", [2, 3, 4])
def test_group_policy_processes_as_ranks(procs):
    """The same policy code and scenario with every rank a PROCESS (fork) over Unix socket pairs: every message is framed
    with (operation, sequence number, size), so ranks whose collectives get out of step or disagree on a size fail at once
    -- what the in-process harness, whose ranks share one address space and one barrier, cannot see."""
    _build()
    r = subprocess.run([EXE, "--procs", str(procs)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "P = %d processes ok" % procs in r.stdout and "group_policy_test: ok" in r.stdout
