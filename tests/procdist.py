"""W ranks as W PROCESSES on one GPU box (tests only): the esp_comm_t callback table of esp_group_create_comm over Unix
sockets, device ranges staged through the host.  What tests/threaddist.py::ThreadComm does for ranks that are threads of
one process, between processes: every rank has its own HIP context, its own library instance and its own esp_group; the
only thing they share is the transport -- every message framed with (operation, sequence number, size), so ranks whose
collectives get out of step fail at once instead of exchanging the wrong bytes.

run_processes(): the supervising launcher (first failure or the deadline ends the other ranks; nothing is left running)."""
import os
import struct
import subprocess
import sys
import time
from multiprocessing.connection import Client, Listener

HDR = struct.Struct("<IIQq")   # operation (1 all-gather, 2 all-to-all-v), sender, sequence number, payload bytes


class ProcComm:
    def __init__(self, rank, world, sockdir, lib_module):
        self.me, self.W, self.L = rank, world, lib_module
        self.seq = 0
        self.conn = [None] * world
        self._keep = []
        self.errors = []
        lst = Listener(os.path.join(sockdir, "l%d" % rank), family="AF_UNIX")
        open(os.path.join(sockdir, "up%d" % rank), "w").close()
        for q in range(rank):                                  # lower ranks listen already or will soon
            path = os.path.join(sockdir, "l%d" % q)
            t0 = time.time()
            while not os.path.exists(os.path.join(sockdir, "up%d" % q)):
                if time.time() - t0 > 120:
                    raise RuntimeError("rank %d never came up" % q)
                time.sleep(0.01)
            c = Client(path, family="AF_UNIX")
            c.send_bytes(struct.pack("<I", rank))
            self.conn[q] = c
        for _ in range(rank + 1, world):
            c = lst.accept()
            q = struct.unpack("<I", c.recv_bytes())[0]
            self.conn[q] = c
        lst.close()

    def _pair(self, q, op, payload, expect_bytes):
        """Exchange one framed message with rank q (the lower rank of the pair sends first)."""
        mine = HDR.pack(op, self.me, self.seq, len(payload)) + payload

        def out():
            self.conn[q].send_bytes(mine)

        def inp():
            msg = self.conn[q].recv_bytes()
            o, frm, seq, nb = HDR.unpack_from(msg)
            if (o, frm, seq) != (op, q, self.seq):
                raise RuntimeError("rank %d expects operation %d number %d from rank %d, got operation %d number %d from rank %d: "
                                   "the ranks' collectives are out of step" % (self.me, op, self.seq, q, o, seq, frm))
            if nb != expect_bytes or len(msg) != HDR.size + nb:
                raise RuntimeError("rank %d expects %d bytes from rank %d, which sends %d" % (self.me, expect_bytes, q, nb))
            return msg[HDR.size:]

        if self.me < q:
            out()
            return inp()
        got = inp()
        out()
        return got

    def table(self, handle_getter):
        """handle_getter(): this rank's esp_handle pointer, known once the matrix exists."""
        import numpy as np
        import torch
        from extendablesparse_devview import view_u8
        L = self.L
        lib = L.load()

        def allgather(ctx, send, count, recv):
            try:
                mine = struct.pack("<%dq" % count, *[send[i] for i in range(count)])
                for q in range(self.W):
                    got = mine if q == self.me else self._pair(q, 1, mine, 8 * count)
                    vals = struct.unpack("<%dq" % count, got)
                    for i in range(count):
                        recv[q * count + i] = vals[i]
                self.seq += 1
                return 0
            except BaseException as e:  # noqa: BLE001 -- reported by the worker
                self.errors.append(e)
                return -3

        def alltoallv(ctx, send, send_bytes, recv, recv_bytes, stream):
            try:
                lib.esp_synchronize(handle_getter())     # the send ranges are complete
                for q in range(self.W):
                    if q == self.me:
                        continue
                    nb, rb = send_bytes[q], recv_bytes[q]
                    payload = view_u8(torch, send[q], nb).cpu().numpy().tobytes() if nb else b""
                    got = self._pair(q, 2, payload, rb)
                    if rb:
                        view_u8(torch, recv[q], rb).copy_(torch.from_numpy(np.frombuffer(got, np.uint8).copy()))
                torch.cuda.synchronize()
                self.seq += 1
                return 0
            except BaseException as e:  # noqa: BLE001
                self.errors.append(e)
                return -3

        t = L.esp_comm_t(None, L.ALLGATHER_FN(allgather), L.ALLTOALLV_FN(alltoallv))
        self._keep.append(t)
        return t


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_processes(cmds, envs, timeout=900.0):
    """Start one process per rank and supervise ALL of them: the first non-zero exit (or the deadline) ends the others,
    nothing is left running.  Returns [(returncode, stdout, stderr)] per rank."""
    procs = [subprocess.Popen(c, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c, e in zip(cmds, envs)]
    t0 = time.time()
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes) or time.time() - t0 > timeout:
                time.sleep(0.5)   # (a rank that failed may have taken the others down with it already)
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=10)
        except Exception:
            o, e = "", "(no output: the rank had to be killed)"
        outs.append((p.returncode, o, e))
    return outs


WORKER = r'''
import json, os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.environ["ESP_ROOT"])
sys.path.insert(0, os.path.join(os.environ["ESP_ROOT"], "tests"))
from esparse_loader import load
from procdist import ProcComm
esp = load()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
sockdir, deal = os.environ["ESP_SOCKDIR"], os.environ["ESP_DEAL"]
d = np.load(os.path.join(sockdir, "in%d.npz" % rank))
N, rounds = int(d["N"]), int(d["rounds"])
comm = ProcComm(rank, world, sockdir, esp._lib)
holder = {}
A = esp.GroupShardedMatrix(N, N, nranks=world, rank=rank, comm=comm.table(lambda: holder["A"].local._d.h))
holder["A"] = A
hist = []
for rnd in range(rounds):
    if deal == "slab":
        nx, ny, nzg, nodes, seed = [int(x) for x in d["slab"][rnd]]
        A.local.generate_fdrand_range(nx, ny, nzg, rank * nodes, (rank + 1) * nodes, seed=seed, rand_mode=1)
    else:
        A.local.append(0, d["I%d" % rnd], d["J%d" % rnd], d["V%d" % rnd], kinds=d["K%d" % rnd])
    A.flush()
    if comm.errors:
        raise comm.errors[0]
    hist.append((A.last_exchange, A.local.debug_last_partition(), A.local.debug_last_shard_source()))
total = A.nnz()
c0, c1, cp, rv, nz = A.local_slice()
np.savez(os.path.join(sockdir, "out%d.npz" % rank), c0=c0, c1=c1, cp=cp, rv=rv, nz=nz, total=total)
print(json.dumps({"hist": hist, "sent": int(A.sent_off_rank), "seq": comm.seq}), flush=True)
'''
