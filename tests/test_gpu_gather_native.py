"""-m gpu: the native gather on the device side.  A test box has one GPU, and RCCL does not put two ranks on one device, so what runs here is
(a) the RCCL transport at world size 1 -- librccl loaded at run time, the unique id, communicator, stream and staging buffers, ncclAllGather and
ncclAllReduce of one rank -- and (b) two PRODUCT processes (HIP kernels) on GPU 0 joined by the socket transport: the C ranks path of
csrc/host/cli.c end to end (sharding, emit hook, rounds, every rank writing its chunks at their offsets / records through rank 0), same SAM as
one process.  The protocol at world sizes 2-4 and the multi-process CPU runs are in tests/test_gather_native.py."""
import ctypes as C
import os
import subprocess
import pytest
import simdata
from biscuit_amd import _lib as B
from test_gather_native import Transport, strip_pg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "biscuit_amd", "biscuit_align")


def test_rccl_transport_one_rank(tmp_path):
    L = B.lib()
    tg, tr = Transport(), Transport()
    L.bsx_transport_rccl.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_void_p]
    B.check(L.bsx_transport_rccl(0, 1, 0, str(tmp_path / "ids").encode(), C.byref(tg), C.byref(tr)), "bsx_transport_rccl")
    assert tg.rank == 0 and tg.world == 1 and tg.ctx and tr.ctx
    ag = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64))(tg.all_gather)
    ar = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int)(tr.all_reduce_sum)
    for rnd in range(5):
        mine = (C.c_int64 * 3)(1, 123456789012 + rnd, -7)
        out = (C.c_int64 * 3)()
        assert ag(tg.ctx, mine, 3, out) == 0 and list(out) == list(mine)
        n = 10001   # a chunk's insert-size histogram: 2 * max_ins + 1 counters
        h = (C.c_int64 * n)(*range(n))
        assert ar(tr.ctx, h, n) == 0 and list(h) == list(range(n))
    for t in (tg, tr):
        C.CFUNCTYPE(None, C.c_void_p)(t.close)(t.ctx)


@pytest.mark.parametrize("mode", ["chunks", "pairs", "via_rank0"])
def test_two_product_processes_over_sockets(tmp_path, mode):
    from biscuit_amd.api import Index
    d = str(tmp_path)
    contigs = simdata.make_genome(400000, seed=9, n_contigs=2)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    pairs = simdata.make_pairs(contigs, 4001, 150, 3, sub=0.01, indel=0.004, pbat_frac=0.2)
    simdata.write_fastq(d + "/r1.fq", [(n, a) for n, a, b in pairs])
    simdata.write_fastq(d + "/r2.fq", [(n, b) for n, a, b in pairs])
    args = ["-@", "2", "g", "r1.fq", "r2.fq"]
    base = dict(os.environ, BSX_CHUNK_SIZE="100000", BSX_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "BSX_OUT", "BSX_GATHER_ID", "BSX_TUNE"):
        base.pop(k, None)
    one = subprocess.run([HIP] + args, cwd=d, env=base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    tune = ["gather_transport=socket"] + (["shard_pairs=1"] if mode == "pairs" else []) + (["gather_via_rank0=1"] if mode == "via_rank0" else [])
    procs = []
    for r in range(2):
        env = dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", BSX_GATHER_ID=d + "/rdv", BSX_TUNE=",".join(tune), BSX_OUT=d + "/two.sam")
        procs.append(subprocess.Popen([HIP] + args, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=1200) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][1].decode()[-3000:])
    a, b = strip_pg(one.stdout), strip_pg(open(d + "/two.sam", "rb").read())
    assert a.count(b"\n") > 8000 and a == b, mode
