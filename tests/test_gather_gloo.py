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


def test_direct_output_needs_one_node_and_a_regular_file(tmp_path):
    """every rank pwrite()s its own chunks only when the ranks share a node and the target is seekable"""
    from biscuit_amd.gather import direct_output_ok
    env = {"LOCAL_WORLD_SIZE": "2"}
    f = tmp_path / "out.sam"
    assert direct_output_ok(str(f), 2, env)                       # will be created: a regular file
    f.write_bytes(b"x")
    assert direct_output_ok(str(f), 2, env)
    assert not direct_output_ok(str(f), 4, env)                   # --nnodes > 1: the ranks do not see one file system
    assert not direct_output_ok(str(f), 2, {})                    # launched by something that does not say
    assert not direct_output_ok("/dev/null", 2, env)
    os.mkfifo(str(tmp_path / "fifo"))
    assert not direct_output_ok(str(tmp_path / "fifo"), 2, env)   # pwrite -> ESPIPE
    assert not direct_output_ok(str(tmp_path / "no_such_dir" / "out.sam"), 2, env)


def _direct_worker(rank, world, n_chunks, init, path, outdir):
    from biscuit_amd.gather import ChunkGather
    dist.init_process_group("gloo", init_method="file://" + init, rank=rank, world_size=world)
    G = ChunkGather(rank, world, torch.device("cpu"), None, max_pending=2, direct_path=path)
    G.header = b"@HD\tVN:1.5\n"

    def produce():
        for k in range(rank, n_chunks, world):
            G.submit(k, _blob(k, False))
        G.close()
    th = threading.Thread(target=produce)
    th.start()
    seen = G.run()   # must return on every rank even when this rank (or another) cannot write
    th.join()
    open(os.path.join(outdir, "r%d" % rank), "w").write("%d %d" % (seen, 1 if G.failed else 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bad", [False, True])
def test_direct_form_survives_a_rank_that_cannot_write(bad):
    """the direct form: the ranks' chunks at their offsets; with a target nobody can open, every rank still finishes all rounds
    (no rank is left inside all_gather) and reports the failure"""
    d = tempfile.mkdtemp(prefix="bsx_direct_")
    path = os.path.join(d, "missing_dir", "out.sam") if bad else os.path.join(d, "out.sam")
    mp.spawn(_direct_worker, args=(2, 7, os.path.join(d, "init"), path, d), nprocs=2, join=True)
    res = [open(os.path.join(d, "r%d" % r)).read().split() for r in range(2)]
    assert [int(x[0]) for x in res] == [7, 7]
    assert [int(x[1]) for x in res] == ([1, 1] if bad else [0, 0])
    if not bad:
        want = b"@HD\tVN:1.5\n" + b"".join(_blob(k, False).tobytes() for k in range(7))
        assert open(path, "rb").read() == want
