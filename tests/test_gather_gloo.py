"""biscuit_amd/gather.py over gloo (world 2 and 3, CPU): rounds, uneven ends, empty chunks, payloads larger than
the staging buffer's first size, back-pressure on the producer, a rank that produces nothing."""
import os
import tempfile
import threading
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _blob(k, big):
    rng = np.random.default_rng(1000 + k)
    n = int(rng.integers(0, 3 << 20 if big else 5000))
    if k % 7 == 3:
        n = 0
    return rng.integers(0, 256, size=n, dtype=np.uint8)


def _worker(rank, world, n_chunks, big, init, outdir):
    from biscuit_amd.gather import ChunkGather
    dist.init_process_group("gloo", init_method="file://" + init, rank=rank, world_size=world)
    got = []
    G = ChunkGather(rank, world, torch.device("cpu"), lambda k, buf: got.append((k, bytes(buf))), max_pending=2)

    def produce():
        for k in range(rank, n_chunks, world):
            G.submit(k, _blob(k, big))
        G.close()
    G.warm(1 << 16 if n_chunks % 2 == 0 else 1)   # the connection-making round before the data (collective; any size)
    th = threading.Thread(target=produce)
    th.start()
    seen = G.run()
    th.join()
    if rank == 0:
        assert [k for k, _ in got] == list(range(n_chunks))
        for k, b in got:
            assert b == _blob(k, big).tobytes(), "chunk %d differs" % k
        open(os.path.join(outdir, "ok"), "w").write("%d %d" % (seen, G.bytes_moved))
    assert seen == n_chunks
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks,big", [(2, 7, True), (2, 8, False), (3, 10, False), (3, 1, False), (2, 0, False)])
def test_streaming_gather(world, n_chunks, big):
    d = tempfile.mkdtemp(prefix="bsx_gather_")
    mp.spawn(_worker, args=(world, n_chunks, big, os.path.join(d, "init"), d), nprocs=world, join=True)
    seen, moved = open(os.path.join(d, "ok")).read().split()
    assert int(seen) == n_chunks
    expect = sum(len(_blob(k, big)) for k in range(n_chunks) if k % world != 0)
    assert int(moved) == expect
