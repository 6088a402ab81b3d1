"""esp_group_* with every rank a PROCESS of its own, on ONE GPU: each rank has its own HIP context, its own instance of
libesparse_hip.so, its own esp_group; the transport is a host callback table (esp_group_create_comm) over Unix sockets
(tests/procdist.py: device ranges staged through the host, every message framed with operation / sequence number / size).
The in-process harness (ranks as threads, one barrier, one address space) cannot show collectives that get out of step
between processes; a multi-GPU node was never available to this build -- this is the closest a one-GPU box gets to
`bench.py --gpus N`: the stitched CSC must equal, bit for bit, ONE oracle buffer fed the ranks' streams in rank order."""
import json
import os
import sys

import numpy as np
import pytest

from refmodel import assert_csc_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UPDATE, RAW = 1, 2


def _run(esp, orc, tmp_path, world, deal, n=44, rounds=3):
    from procdist import WORKER, run_processes
    nx = ny = n
    nzg = n * world
    N = nx * ny * nzg
    nodes = nx * ny * n
    seeds = [77, 78, 79][:rounds]
    streams = [orc.fdrand_stream(nx, ny, nzg, rand_mode=1, seed=s) for s in seeds]
    E = len(streams[0][0])
    kinds = np.where(np.arange(E) % 5 == 0, RAW, UPDATE).astype(np.uint8)
    if deal == "slab":
        l = np.arange(N)
        i, j, k = l % nx + 1, (l // nx) % ny + 1, l // (nx * ny) + 1
        per = (4 * (i < nx) + ((i == 1) | (i == nx)) + 4 * (j < ny) + ((j == 1) | (j == ny))
               + 4 * (k < nzg) + ((k == 1) | (k == nzg)))
        off = np.concatenate([[0], np.cumsum(per)])
        sel = [slice(int(off[r * nodes]), int(off[(r + 1) * nodes])) for r in range(world)]
        kinds[:] = UPDATE
    else:
        chunk = np.arange(E) // 4096
        sel = [(chunk % world) == r for r in range(world)]
    perm1 = np.random.default_rng(9).permutation(int(np.count_nonzero(sel[1])) if deal != "slab" else 1)

    def rank_stream(rank, rnd):
        I, J, V = streams[rnd]
        Ii, Jj, Vv, kk = I[sel[rank]], J[sel[rank]], V[sel[rank]], kinds[sel[rank]]
        if deal == "shuffled_rank1" and rank == 1:
            Ii, Jj, Vv, kk = Ii[perm1], Jj[perm1], Vv[perm1], kk[perm1]
        return Ii, Jj, Vv, kk

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmds, envs = [], []
    for r in range(world):
        data = {"N": N, "rounds": rounds, "slab": np.array([[nx, ny, nzg, nodes, s] for s in seeds], np.int64)}
        if deal != "slab":
            for rnd in range(rounds):
                Ii, Jj, Vv, kk = rank_stream(r, rnd)
                data.update({"I%d" % rnd: Ii, "J%d" % rnd: Jj, "V%d" % rnd: Vv, "K%d" % rnd: kk})
        np.savez(tmp_path / ("in%d.npz" % r), **data)
        cmds.append([sys.executable, str(script)])
        envs.append(dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), ESP_ROOT=ROOT, ESP_SOCKDIR=str(tmp_path), ESP_DEAL=deal))
    outs = run_processes(cmds, envs, timeout=600)
    assert all(o[0] == 0 for o in outs), [(o[0], o[2][-1500:]) for o in outs]
    reports = [json.loads([ln for ln in o[1].splitlines() if ln.startswith("{")][-1]) for o in outs]
    O = orc.ExtendableSparseMatrix(N, N)
    for rnd in range(rounds):
        for r in range(world):
            Ii, Jj, Vv, kk = rank_stream(r, rnd)
            O.apply(kk, Ii, Jj, Vv)
        O.flush()
    pieces, totals = [], []
    for r in range(world):
        d = np.load(tmp_path / ("out%d.npz" % r))
        pieces.append((int(d["c0"]), int(d["c1"]), d["cp"], d["rv"], d["nz"]))
        totals.append(int(d["total"]))
    G = esp.GroupShardedMatrix.stitch(N, N, pieces, totals[0])
    assert_csc_equal(G.arrays(), O.arrays())
    assert all(t == O.nnz() for t in totals)
    assert len({rep["seq"] for rep in reports}) == 1            # every rank issued the same number of collectives
    return reports


@pytest.mark.gpu
@pytest.mark.parametrize("world,deal", [(2, "slab"), (3, "slab"), (2, "scrambled")])
def test_group_api_ranks_as_processes(esp, orc, tmp_path, world, deal):
    reports = _run(esp, orc, tmp_path, world, deal)
    for rep in reports:
        assert [h[0] for h in rep["hist"]] == ["partitioned"] * 3 and all(h[1] == 7 for h in rep["hist"]), rep
        if deal == "slab":
            # (from the second assembly on the producers partition by (owner, digit) themselves: esp_shard_plan)
            assert [h[2] for h in rep["hist"]] == [1, 2, 2], rep
    # (a slab's cross-slab pairs are issued by the lower slab: every rank but the last sends entries off rank)
    assert all(rep["sent"] > 0 for rep in reports[:-1])


@pytest.mark.gpu
def test_group_api_processes_fall_back_by_consensus(esp, orc, tmp_path):
    """Rank 1's stream is shuffled: every rank takes the in-place exchange for that flush -- a decision all processes must
    reach from all-gathered data alone -- and backs off the same number of flushes."""
    reports = _run(esp, orc, tmp_path, 2, "shuffled_rank1")
    kinds = [[h[0] for h in rep["hist"]] for rep in reports]
    assert kinds[0] == kinds[1] and kinds[0][0] == "inplace", kinds
