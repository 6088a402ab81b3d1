"""torch.distributed stand-in for W ranks that live in W threads of ONE process (tests only).

The GPU box of this build has one GPU: the multi-rank code path of sharded.py (partitioned exchange,
piece tables with several sources, consensus fall-back) is exercised there by running the ranks as
threads, every rank with its own library handle on the same device.  Collectives are deposit /
barrier / copy / barrier."""
import threading

import torch


class ThreadDist:
    class ReduceOp:
        MAX = "max"
        SUM = "sum"

    def __init__(self, world):
        self.W = world
        self._bar = threading.Barrier(world)
        self._box = [None] * world
        self._tl = threading.local()

    # -- rank bookkeeping
    def bind(self, rank):
        self._tl.rank = rank

    def get_rank(self, group=None):
        return self._tl.rank

    def get_world_size(self, group=None):
        return self.W

    def _deposit(self, obj):
        self._box[self.get_rank()] = obj
        self._bar.wait()
        return list(self._box)

    def _done(self):
        self._bar.wait()

    # -- collectives used by sharded.py
    def barrier(self, group=None):
        self._bar.wait()

    def all_gather(self, outs, t, group=None):
        allv = self._deposit(t)
        for q in range(self.W):
            outs[q].copy_(allv[q])
        torch.cuda.synchronize() if t.is_cuda else None
        self._done()

    def all_reduce(self, t, op=None, group=None):
        allv = self._deposit(t.clone())
        st = torch.stack([x.to(t.device) for x in allv])
        t.copy_(st.max(0).values if op == "max" else st.sum(0))
        torch.cuda.synchronize() if t.is_cuda else None
        self._done()

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None, group=None):
        me = self.get_rank()
        if in_splits is None:
            c = inp.numel() // self.W
            in_splits = [c] * self.W
        allv = self._deposit((inp, list(in_splits)))
        pos = 0
        for q in range(self.W):
            src, sp = allv[q]
            o = sum(sp[:me])
            n = sp[me]
            if out_splits is not None:
                assert out_splits[q] == n, (out_splits, q, n)
            out[pos:pos + n].copy_(src[o:o + n])
            pos += n
        torch.cuda.synchronize() if out.is_cuda else None
        self._done()

    def gather_object(self, obj, out, dst=0, group=None):
        allv = self._deposit(obj)
        if self.get_rank() == dst:
            for q in range(self.W):
                out[q] = allv[q]
        self._done()


def run_ranks(world, fn):
    """fn(rank, dist) in `world` threads; re-raises the first failure."""
    dist = ThreadDist(world)
    errs = [None] * world
    outs = [None] * world

    def body(r):
        dist.bind(r)
        try:
            outs[r] = fn(r, dist)
        except BaseException as e:  # noqa: BLE001 -- reported below
            errs[r] = e
            dist._bar.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    real = [e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real:
        raise real[0]
    for e in errs:
        if e is not None:
            raise e
    return outs


class ThreadComm:
    """esp_comm_t callback tables (esp_group_create_comm) for W ranks that live in W threads of one process:
    the C group API's exchange driven with a transport of the host.  deposit / barrier / copy / barrier."""

    def __init__(self, world, lib_module):
        self.W = world
        self.L = lib_module            # extendablesparse.jl_amd._lib
        self._bar = threading.Barrier(world)
        self._box = [None] * world
        self._keep = []
        self.errors = []

    def table(self, rank, handle_getter):
        """handle_getter(): the rank's esp_handle pointer (c_void_p), known once the matrix exists."""
        import ctypes as C
        L = self.L
        lib = L.load()

        def wrap(ptr, nbytes):
            from extendablesparse_devview import view_u8
            return view_u8(torch, ptr, nbytes)

        def allgather(ctx, send, count, recv):
            try:
                self._box[rank] = [send[i] for i in range(count)]
                self._bar.wait()
                for q in range(self.W):
                    for i in range(count):
                        recv[q * count + i] = self._box[q][i]
                self._bar.wait()
                return 0
            except BaseException as e:  # noqa: BLE001
                self.errors.append(e)
                self._bar.abort()
                return -3

        def alltoallv(ctx, send, send_bytes, recv, recv_bytes, stream):
            try:
                lib.esp_synchronize(handle_getter())     # the send ranges are complete
                self._box[rank] = ([send[q] for q in range(self.W)], [send_bytes[q] for q in range(self.W)])
                self._bar.wait()
                for q in range(self.W):
                    if q == rank:
                        continue
                    src, nb = self._box[q][0][rank], self._box[q][1][rank]
                    assert nb == recv_bytes[q], (rank, q, nb, recv_bytes[q])
                    if nb:
                        wrap(recv[q], nb).copy_(wrap(src, nb))
                torch.cuda.synchronize()
                self._bar.wait()
                return 0
            except BaseException as e:  # noqa: BLE001
                self.errors.append(e)
                self._bar.abort()
                return -3

        t = L.esp_comm_t(None, L.ALLGATHER_FN(allgather), L.ALLTOALLV_FN(alltoallv))
        self._keep.append(t)
        return t
