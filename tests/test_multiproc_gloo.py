"""CPU, world_size 2 over gloo: the chunk-sharded multi-process path (biscuit_amd/multi_gpu.py) produces
exactly the single-process SAM.  Uses the CPU restatement of the kernels (no GPU here); the sharding,
offset bookkeeping and ordered gather are the same code the GPU launcher runs."""
import os
import subprocess
import sys
import pytest
import simdata
from biscuit_amd.api import Index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_pg(b):
    return b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))


@pytest.mark.parametrize("pe", [True, False])
def test_two_ranks_equal_one(tmp_path, pe):
    d = str(tmp_path)
    contigs = simdata.make_genome(200000, seed=5, n_contigs=2)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    # ~3.6 Mbp of reads with -@ 1 and a 1 Mbp chunk => several chunks (chunk size is not a CLI option in the
    # reference either; the test shrinks it through the environment)
    pairs = simdata.make_pairs(contigs, 3000, 100, 3, frag=(150, 300), sub=0.01, indel=0.003)
    simdata.write_fastq(d + "/r1.fq", [(n, a) for n, a, b in pairs])
    simdata.write_fastq(d + "/r2.fq", [(n, b) for n, a, b in pairs])
    files = ["r1.fq", "r2.fq"] if pe else ["r1.fq"]
    env = dict(os.environ, BSX_CHUNK_SIZE="100000", PYTHONPATH=ROOT)
    one = subprocess.run([os.path.join(ROOT, "oracle", "oracle_align"), "-@", "1", "g"] + files, cwd=d, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    assert one.stderr.count(b"sequences (") >= 3     # really several chunks
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29531" if pe else "29532", os.path.join(ROOT, "tests", "multi_entry_cpu.py"), "--out", d + "/two.sam", "--", "-@", "1", "g"] + files,
                         cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert two.returncode == 0, two.stderr.decode()[-3000:]
    a, b = strip_pg(one.stdout), strip_pg(open(d + "/two.sam", "rb").read())
    assert a.count(b"\n") > 3000
    assert a == b
    # plain files: each rank found its chunks through the boundary scan and parsed nothing else
    assert two.stderr.count(b"this rank parses its own chunks only") == 2, two.stderr.decode()[-2000:]
    if pe:
        # (the run above wrote its file in the direct form -- every rank its own chunks at their offsets; this one gathers to rank 0)
        # compressed input has no offsets to seek to: every rank inflates everything, parses its own chunks and walks the other rank's
        # without building records (bsx_fq_skip_chunk); same SAM
        import gzip
        import shutil
        for f in files:
            with open(d + "/" + f, "rb") as fi, gzip.open(d + "/" + f + ".gz", "wb") as fo:
                shutil.copyfileobj(fi, fo)
        gz = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                             "--master-port", "29533", os.path.join(ROOT, "tests", "multi_entry_cpu.py"), "--out", d + "/gz.sam", "--via-rank0", "--", "-@", "1", "g"] + [f + ".gz" for f in files],
                            cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert gz.returncode == 0, gz.stderr.decode()[-3000:]
        assert strip_pg(open(d + "/gz.sam", "rb").read()) == a
        assert gz.stderr.count(b"walks the others without building records") == 2


@pytest.mark.parametrize("pe,n_ranks", [(True, 2), (False, 2), (True, 3)])
def test_ranks_sharing_each_chunk_equal_one(tmp_path, pe, n_ranks):
    """--shard pairs: every rank aligns its slice of every chunk, the insert-size histograms are added over the ranks
    (bsx_pes_hist_hook), the slices leave in rank order.  Same SAM as one process -- with several chunks, with ONE chunk (fewer chunks
    than ranks: what chunk sharding cannot spread), and with a last chunk of fewer pairs than ranks (empty slices)."""
    d = str(tmp_path)
    contigs = simdata.make_genome(200000, seed=6, n_contigs=2)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    port = 29560 + (4 if pe else 0) + n_ranks * 8
    for tag, n_pairs, chunk in (("multi", 2001, "100000"), ("single", 900, "10000000")):
        # 2001 pairs of 2 x 100 at 100 kbp per chunk: 500 pairs a chunk, the last chunk is ONE pair (empty slices on the other ranks)
        pairs = simdata.make_pairs(contigs, n_pairs, 100, 3, frag=(150, 300), sub=0.01, indel=0.003)
        simdata.write_fastq(d + "/%s_1.fq" % tag, [(n, a) for n, a, b in pairs])
        simdata.write_fastq(d + "/%s_2.fq" % tag, [(n, b) for n, a, b in pairs])
        files = ["%s_1.fq" % tag, "%s_2.fq" % tag] if pe else ["%s_1.fq" % tag]
        env = dict(os.environ, BSX_CHUNK_SIZE=chunk, PYTHONPATH=ROOT)
        one = subprocess.run([os.path.join(ROOT, "oracle", "oracle_align"), "-@", "1", "g"] + files, cwd=d, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert one.returncode == 0, one.stderr.decode()[-2000:]
        n_chunks = one.stderr.count(b"sequences (")
        assert (n_chunks >= 3) if tag == "multi" else (n_chunks == 1)
        port += 1
        many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
                               "--master-port", str(port), os.path.join(ROOT, "tests", "multi_entry_cpu.py"), "--out", d + "/many.sam", "--shard", "pairs"]
                              + (["--via-rank0"] if n_ranks == 3 else []) + ["--", "-@", "1", "g"] + files, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert many.returncode == 0, many.stderr.decode()[-3000:]
        a, b = strip_pg(one.stdout), strip_pg(open(d + "/many.sam", "rb").read())
        assert a.count(b"\n") > (800 if pe else 400)
        assert a == b, tag
        # every rank read every chunk (nobody skipped), and with pairs every rank took part in every chunk's statistics
        assert many.stderr.count(b"sequences (") == n_chunks * n_ranks, many.stderr.decode()[-2000:]
