"""bench.py --gpus N: the parent only launches N rank processes (it never touches the GPU), relays rank 0's line
and refuses to run when the node has fewer GPUs than asked for.  CPU-only: the ranks meet over gloo (probe hook)."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("esp_bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Args:
    def __init__(self, gpus):
        self.gpus = gpus


def test_launcher_refuses_when_the_node_has_fewer_gpus(monkeypatch, capfd):
    bench = _bench()
    monkeypatch.setattr(bench, "gpu_count_without_hip", lambda: 1)   # (whatever this host has: a GPU box has /sys/class/kfd)
    rc = bench.launch(_Args(2), ["--gpus", "2"])
    assert rc != 0
    assert "refusing" in capfd.readouterr().err


@pytest.mark.parametrize("n", [2, 3])
def test_launcher_starts_n_ranks(monkeypatch, capfd, n):
    bench = _bench()
    monkeypatch.setattr(bench, "gpu_count_without_hip", lambda: 8)
    monkeypatch.setenv("ESP_BENCH_LAUNCH_PROBE", "1")
    monkeypatch.setenv("MASTER_PORT", str(29600 + n))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.launch(_Args(n), ["--gpus", str(n), "--steps", "1"])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and out, out
    d = json.loads(out[-1])
    assert d["probe"] and d["n_gpus"] == n and d["rank_sum"] == n * (n - 1) / 2
    assert d["local_rank"] == 0 and d["master"] == "127.0.0.1"


def test_launcher_ends_the_job_when_a_rank_dies(monkeypatch, capfd):
    """A rank other than 0 that exits at start-up must not leave rank 0 waiting in the rendezvous: the launcher ends the
    others, returns the failing rank's code and shows its stderr."""
    import time
    bench = _bench()
    monkeypatch.setattr(bench, "gpu_count_without_hip", lambda: 8)
    monkeypatch.setenv("ESP_BENCH_LAUNCH_PROBE", "1")
    monkeypatch.setenv("ESP_BENCH_PROBE_FAIL_RANK", "1")
    monkeypatch.setenv("MASTER_PORT", "29611")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    t0 = time.time()
    rc = bench.launch(_Args(2), ["--gpus", "2", "--steps", "1"])
    err = capfd.readouterr().err
    assert rc == 7 and time.time() - t0 < 120
    assert "rank 1 exited with code 7" in err and "fails on purpose" in err


def test_gpu_count_follows_the_visible_devices_lists(monkeypatch, tmp_path):
    """The sysfs count is clamped by the *_VISIBLE_DEVICES lists the ranks inherit; without sysfs torch counts."""
    bench = _bench()
    real_listdir, real_open = os.listdir, open
    base = "/sys/class/kfd/kfd/topology/nodes"
    nodes = {"0": "simd_count 0\n", "1": "simd_count 1024\n", "2": "simd_count 1024\n", "3": "simd_count 1024\n"}
    for k, v in nodes.items():
        (tmp_path / k).mkdir()
        (tmp_path / k / "properties").write_text("cpu_cores_count 0\n" + v)
    monkeypatch.setattr(bench.os, "listdir", lambda p: list(nodes) if p == base else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace(base, str(tmp_path)), *a, **k))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.gpu_count_without_hip() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.gpu_count_without_hip() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.gpu_count_without_hip() == 1
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert bench.gpu_count_without_hip() == 0


def test_world_size_mismatch_is_an_error():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], env=env,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
