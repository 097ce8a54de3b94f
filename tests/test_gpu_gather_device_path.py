"""GPU: the device branch of biscuit_amd/gather.py (staging tensors in HBM, pinned host buffers on both sides) with the
process group replaced by an in-process stand-in: two ranks as two threads on cuda:0, payloads moved by tensor copies
where RCCL would move them.  A one-GPU box cannot run two RCCL ranks; the CPU tests (gloo) cover the protocol, this
covers the copies the CPU tests never execute."""
import queue
import threading

import numpy as np
import pytest
import torch

import biscuit_amd.gather as gather_mod

pytestmark = pytest.mark.gpu


class _Group:
    """all_gather / send / recv between threads; the calling thread's rank is kept in a thread-local"""

    def __init__(self, world):
        self.world = world
        self.tl = threading.local()
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}

    def all_gather(self, out, t):
        self.slots[self.tl.rank] = t.clone()
        self.bar.wait()
        for i in range(self.world):
            out[i].copy_(self.slots[i])
        self.bar.wait()

    def send(self, buf, dst):
        assert buf.is_cuda
        self.q[(self.tl.rank, dst)].put(buf.clone())

    def recv(self, buf, src):
        assert buf.is_cuda
        buf.copy_(self.q[(src, self.tl.rank)].get())

    # rank 0 posts a round's receives together (batch_isend_irecv over P2POp(irecv, ...)): here each becomes a deferred copy
    def irecv(self, buf, src):
        raise AssertionError("only through batch_isend_irecv")

    class _Op:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    def P2POp(self, op, tensor, peer):
        assert op == self.irecv and tensor.is_cuda
        return self._Op(op, tensor, peer)

    class _Req:
        def __init__(self, fn):
            self.fn = fn

        def wait(self):
            self.fn()

    def batch_isend_irecv(self, ops):
        me = self.tl.rank
        return [self._Req(lambda o=o: o.tensor.copy_(self.q[(o.peer, me)].get())) for o in ops]


@pytest.mark.parametrize("n_chunks,world", [(5, 2), (8, 2), (11, 3)])
def test_device_branch_moves_the_records(monkeypatch, n_chunks, world):
    grp = _Group(world)
    monkeypatch.setattr(gather_mod, "dist", grp)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    sizes = [0, 1, 77, 1 << 20, (3 << 20) + 5, 13, 2 << 20, 4096][:n_chunks] + [9] * max(0, n_chunks - 8)
    chunks = [rng.integers(0, 256, n, dtype=np.uint8) for n in sizes]
    got = []

    def sink(k, buf):
        got.append((k, bytes(buf)))

    gs = [gather_mod.ChunkGather(r, world, dev, sink if r == 0 else None, max_pending=2) for r in range(world)]
    res = [None] * world

    def consume(r):
        grp.tl.rank = r
        torch.cuda.set_device(0)
        res[r] = gs[r].run()

    def produce(r):
        for k in range(r, n_chunks, world):
            gs[r].submit(k, chunks[k].copy())
        gs[r].close()

    th = [threading.Thread(target=consume, args=(r,)) for r in range(world)] + [threading.Thread(target=produce, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert all(x == n_chunks for x in res)
    assert [k for k, _ in got] == list(range(n_chunks))
    for k, b in got:
        assert b == chunks[k].tobytes()
    assert gs[0].bytes_moved == sum(sizes[k] for k in range(n_chunks) if k % world)
