"""The gather in C (csrc/host/gather.c; include/bsx.h: bsx_gather_*, bsx_transport_*): SURVEY 8(e)'s "replicas + gather" without Python in the
data path.  CPU suite:
  * the protocol over the in-process transport (the ranks as threads): rounds, ordering, uneven ends, empty chunks, a rank that produces
    nothing, payloads of megabytes, both forms (records through rank 0 / every rank writes its own chunks at their offsets), a rank that
    cannot write;
  * the socket transport's collectives;
  * the command line as 2 and 3 PROCESSES (RANK / WORLD_SIZE in the environment, Unix sockets between them, the CPU checker's backend):
    the single-process SAM, chunk-sharded and with every rank taking its slice of every chunk.
(-m gpu: the RCCL transport's one-rank paths and two product processes sharing GPU 0 over sockets, tests/test_gpu_gather_native.py.)"""
import ctypes as C
import os
import subprocess
import threading
import numpy as np
import pytest
from biscuit_amd import _lib as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("all_gather", C.c_void_p), ("send", C.c_void_p), ("recv_many", C.c_void_p),
                ("all_reduce_sum", C.c_void_p), ("close", C.c_void_p)]


SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t)
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]


def _api():
    L = B.lib()
    L.bsx_gather_open.argtypes = [C.c_void_p, C.c_char_p, SINK, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.bsx_gather_submit.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t]
    L.bsx_gather_set_header.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.bsx_gather_close_input.argtypes = [C.c_void_p]
    L.bsx_gather_run.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.bsx_gather_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.bsx_gather_free.argtypes = [C.c_void_p]
    L.bsx_gather_free.restype = None
    return L


def _blob(k):
    rng = np.random.default_rng(7000 + k)
    n = 0 if k % 7 == 3 else int(rng.integers(1, 3 << 20 if k % 5 == 0 else 6000))
    return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()


def _close(tr):
    C.CFUNCTYPE(None, C.c_void_p)(tr.close)(tr.ctx)


def _run_ranks(world, n_chunks, direct=None, header=b"", silent_rank=None):
    """every rank a thread: a producer thread submits its chunks, the rank's thread runs the rounds"""
    L = _api()
    trs = (Transport * world)()
    B.check(L.bsx_transport_local(world, trs), "bsx_transport_local")
    got = [[] for _ in range(world)]
    res = [None] * world
    keep = []

    def rank_main(r):
        def sink(ud, chunk, buf, n):
            got[r].append((chunk, C.string_at(buf, n) if n else b""))
        cb = SINK(sink)
        keep.append(cb)
        g = C.c_void_p()
        B.check(L.bsx_gather_open(C.byref(trs[r]), direct.encode() if direct else None, cb, None, 2, C.byref(g)), "open")
        if r == 0 and header:
            L.bsx_gather_set_header(g, header, len(header))

        def produce():
            if r != silent_rank:
                for k in range(r, n_chunks, world):
                    b = _blob(k)
                    p = libc.malloc(max(1, len(b)))
                    C.memmove(p, b, len(b))
                    L.bsx_gather_submit(g, k, p, len(b))
            L.bsx_gather_close_input(g)
        th = threading.Thread(target=produce)
        th.start()
        n = C.c_int64()
        rc = L.bsx_gather_run(g, C.byref(n))
        th.join()
        st = (C.c_int64 * 3)()
        L.bsx_gather_stats(g, st)
        res[r] = (rc, n.value, list(st))
        L.bsx_gather_free(g)
        _close(trs[r])
    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
        assert not t.is_alive(), "a rank is stuck in the rounds"
    return got, res


@pytest.mark.parametrize("world,n_chunks", [(2, 7), (2, 8), (3, 10), (3, 1), (4, 13), (2, 0), (1, 5)])
def test_records_through_rank0(world, n_chunks):
    got, res = _run_ranks(world, n_chunks)
    assert [k for k, _ in got[0]] == list(range(n_chunks))                # rank 0 sees every chunk, in input order
    for k, b in got[0]:
        assert b == _blob(k), k
    for r in range(1, world):
        assert got[r] == []
    assert all(rc == 0 and n == n_chunks for rc, n, _ in res), res        # every rank counted the same chunks and left the rounds
    assert res[0][2][1] == sum(len(_blob(k)) for k in range(n_chunks) if k % world != 0)   # payload bytes that moved = the other ranks' chunks, exactly


def test_a_rank_without_chunks_ends_the_rounds_for_everybody():
    got, res = _run_ranks(3, 9, silent_rank=1)
    # chunk 1 never comes: round 0 is the last one (round-robin dealing: no later chunk can be written in order)
    assert [k for k, _ in got[0]] == [0, 2] and all(rc == 0 and n == 2 for rc, n, _ in res)


@pytest.mark.parametrize("world,n_chunks", [(2, 7), (3, 10), (4, 5)])
def test_every_rank_writes_its_own_chunks(tmp_path, world, n_chunks):
    path = str(tmp_path / "out.sam")
    hdr = b"@HD\tVN:1.5\n@SQ\tSN:c\tLN:9\n"
    got, res = _run_ranks(world, n_chunks, direct=path, header=hdr)
    assert all(g == [] for g in got) and all(rc == 0 and n == n_chunks for rc, n, _ in res)
    assert open(path, "rb").read() == hdr + b"".join(_blob(k) for k in range(n_chunks))
    for r in range(world):
        assert res[r][2][2] == sum(len(_blob(k)) for k in range(r, n_chunks, world))      # each rank wrote its own bytes, nothing moved
        assert res[r][2][1] == 0


def test_a_rank_that_cannot_write_still_finishes_every_round(tmp_path):
    got, res = _run_ranks(3, 8, direct=str(tmp_path / "no_such_dir" / "out.sam"))
    assert all(n == 8 for _, n, _ in res) and all(rc == -3 for rc, _, _ in res), res      # BSX_E_IO everywhere, nobody stuck


def test_direct_needs_one_node_and_a_regular_file(tmp_path):
    L = _api()
    L.bsx_gather_direct_ok.argtypes = [C.c_char_p, C.c_int, C.c_int]
    f = tmp_path / "o.sam"
    assert L.bsx_gather_direct_ok(str(f).encode(), 2, 2) == 1
    f.write_bytes(b"x")
    assert L.bsx_gather_direct_ok(str(f).encode(), 2, 2) == 1
    assert L.bsx_gather_direct_ok(str(f).encode(), 4, 2) == 0          # more ranks than this node holds
    assert L.bsx_gather_direct_ok(b"/dev/null", 2, 2) == 0
    os.mkfifo(str(tmp_path / "fifo"))
    assert L.bsx_gather_direct_ok(str(tmp_path / "fifo").encode(), 2, 2) == 0


def _collectives(trs, r, out):
    ag = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64))(trs[r].all_gather)
    ar = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int)(trs[r].all_reduce_sum)
    world = trs[r].world
    for rnd in range(20):
        mine = (C.c_int64 * 3)(r, rnd, r * 100 + rnd)
        allv = (C.c_int64 * (3 * world))()
        assert ag(trs[r].ctx, mine, 3, allv) == 0
        assert list(allv) == [v for q in range(world) for v in (q, rnd, q * 100 + rnd)]
        h = (C.c_int64 * 5)(*[r + k * rnd for k in range(5)])
        assert ar(trs[r].ctx, h, 5) == 0
        assert list(h) == [sum(q + k * rnd for q in range(world)) for k in range(5)]
    out[r] = True


@pytest.mark.parametrize("world", [2, 4])
def test_local_transport_collectives(world):
    L = _api()
    trs = (Transport * world)()
    B.check(L.bsx_transport_local(world, trs), "local")
    ok = [False] * world
    ths = [threading.Thread(target=_collectives, args=(trs, r, ok)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=60)
    assert all(ok)
    for r in range(world):
        _close(trs[r])


def strip_pg(b):
    return b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))


@pytest.mark.parametrize("world,mode", [(2, "chunks"), (3, "chunks"), (2, "pairs"), (3, "pairs"), (2, "via_rank0"), (2, "stdout")])
def test_processes_over_sockets_equal_one_process(tmp_path, world, mode):
    """`oracle_align` started once per rank with RANK / WORLD_SIZE / LOCAL_RANK: the native ranks path of csrc/host/cli.c over the socket
    transport.  Same SAM as one process -- chunks dealt to the ranks; every rank a slice of every chunk (histograms added over the ranks);
    the records through rank 0 into a file and to its stdout."""
    import simdata
    from biscuit_amd.api import Index
    d = str(tmp_path)
    contigs = simdata.make_genome(200000, seed=5, n_contigs=2)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    pairs = simdata.make_pairs(contigs, 2501, 100, 3, frag=(150, 300), sub=0.01, indel=0.003)   # 2501: the last chunk is one pair (empty slices)
    simdata.write_fastq(d + "/r1.fq", [(n, a) for n, a, b in pairs])
    simdata.write_fastq(d + "/r2.fq", [(n, b) for n, a, b in pairs])
    exe = os.path.join(ROOT, "oracle", "oracle_align")
    args = ["-@", "1", "g", "r1.fq", "r2.fq"]
    base = dict(os.environ, BSX_CHUNK_SIZE="100000")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "BSX_OUT", "BSX_GATHER_ID", "BSX_TUNE"):
        base.pop(k, None)
    one = subprocess.run([exe] + args, cwd=d, env=base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    assert one.stderr.count(b"sequences (") >= 3
    tune = ["gather_transport=socket"] + (["shard_pairs=1"] if mode == "pairs" else []) + (["gather_via_rank0=1"] if mode == "via_rank0" else [])
    procs = []
    for r in range(world):
        env = dict(base, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world), BSX_GATHER_ID=d + "/rdv", BSX_TUNE=",".join(tune))
        if mode != "stdout":
            env["BSX_OUT"] = d + "/many.sam"
        procs.append(subprocess.Popen([exe] + args, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=900) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][1].decode()[-3000:])
    text = outs[0][0] if mode == "stdout" else open(d + "/many.sam", "rb").read()
    for r in range(1, world):
        assert outs[r][0] == b""                                   # only rank 0 writes to stdout
    a, b = strip_pg(one.stdout), strip_pg(text)
    assert a.count(b"\n") > 2500 and a == b, mode
    if mode == "pairs":   # every rank read every chunk
        assert sum(o[1].count(b"sequences (") for o in outs) == one.stderr.count(b"sequences (") * world
