"""-m gpu: the whole `biscuit align` path.  SAM produced by the product (HIP kernels) must equal,
byte for byte (minus @PG), the SAM produced by the same host pipeline over the CPU restatement of
the kernels (oracle/).  Covers BASELINE.json's configs at sizes the CPU path finishes in seconds."""
import os
import subprocess
import numpy as np
import pytest
import simdata

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "biscuit_amd", "biscuit_align")
CPU = os.path.join(ROOT, "oracle", "oracle_align")


def run(exe, args, cwd):
    p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    from biscuit_amd.api import Index
    d = str(tmp_path_factory.mktemp("align"))
    contigs = simdata.make_genome(1000000, seed=21, n_contigs=3)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    # config 1 shape: 2x100 directional pairs; plus a harder 2x150 set (indels, chimeras, PBAT-like, bad mates, Ns)
    p100 = simdata.make_pairs(contigs, 5000, 100, 1, frag=(180, 320), sub=0.005)
    p150 = simdata.make_pairs(contigs, 4000, 150, 2, sub=0.01, indel=0.006, pbat_frac=0.3, chimera_frac=0.06, bad_mate_frac=0.06, n_frac=0.03)
    for tag, pairs in (("a", p100), ("b", p150)):
        simdata.write_fastq(d + "/%s1.fq" % tag, [(n, a) for n, a, b in pairs])
        simdata.write_fastq(d + "/%s2.fq" % tag, [(n, b) for n, a, b in pairs])
    simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, 300, 1000, 5))
    return d


CASES = [
    ("pe100_default", ["-@", "4", "g", "a1.fq", "a2.fq"]),
    ("pe150_default", ["-@", "4", "g", "b1.fq", "b2.fq"]),
    ("pe150_directional", ["-@", "4", "-b", "1", "g", "b1.fq", "b2.fq"]),
    ("pe150_all_softclip", ["-@", "4", "-a", "-Y", "g", "b1.fq", "b2.fq"]),
    ("pe150_norescue_nopair", ["-@", "4", "-S", "-P", "g", "b1.fq", "b2.fq"]),
    ("pe150_fixed_isize_rg", ["-@", "4", "-I", "350,60", "-R", "@RG\\tID:x\\tSM:y", "-C", "g", "b1.fq", "b2.fq"]),
    ("se150_default", ["-@", "4", "g", "b1.fq"]),
    ("se150_daughter", ["-@", "4", "-b", "3", "g", "b2.fq"]),
    ("se150_clip", ["-@", "4", "-J", "AGATCGGAAGAGC", "-z", "10", "-5", "2", "-3", "1", "g", "b1.fq"]),
    ("se150_scoring", ["-@", "4", "-A", "2", "-B", "3", "-O", "5,7", "-E", "2,1", "-L", "4,6", "-T", "40", "-k", "17", "-w", "60", "g", "b1.fq"]),
    ("long_1kb", ["-@", "4", "g", "long.fq"]),
]


@pytest.mark.parametrize("name,args", CASES, ids=[c[0] for c in CASES])
def test_sam_identical(data, name, args):
    want = run(CPU, args, data)
    got = run(HIP, args, data)
    assert got.count(b"\n") > 100
    if got != want:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(gl, wl)):
            assert a == b, "first difference at line %d:\nHIP: %s\nCPU: %s" % (i, a[:600], b[:600])
        assert len(gl) == len(wl)
