"""CPU: the host pipeline end to end (`oracle_align` = biscuit_align's driver over the CPU restatement
of the kernels).  Checks SAM validity against the genome, thread-count independence, option handling
and a committed regression fixture.  NOTE: the fixture under tests/golden/pe_small.sam was produced by
THIS repository's CPU path -- the reference's chaining/pairing/formatting files cannot be built offline
(un-vendored headers), so it guards against regressions, it is not a reference output."""
import os
import subprocess
import numpy as np
import pytest
import simdata
import samcheck
from biscuit_amd.api import Index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "oracle", "oracle_align")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(args, cwd):
    p = subprocess.run([CPU] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return "\n".join(l for l in p.stdout.decode().split("\n") if not l.startswith("@PG"))


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("cpu_align"))
    contigs = simdata.make_genome(300000, seed=21, n_contigs=3)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    pairs = simdata.make_pairs(contigs, 1500, 100, 1, frag=(180, 320), sub=0.005)
    hard = simdata.make_pairs(contigs, 1500, 150, 2, sub=0.01, indel=0.006, pbat_frac=0.3, chimera_frac=0.06, bad_mate_frac=0.06, n_frac=0.03)
    for tag, ps in (("a", pairs), ("b", hard)):
        simdata.write_fastq(d + "/%s1.fq" % tag, [(n, a) for n, a, b in ps])
        simdata.write_fastq(d + "/%s2.fq" % tag, [(n, b) for n, a, b in ps])
    simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, 40, 1000, 5))
    # edge cases: empty-ish, all-N, shorter than the seed length, adaptor tail
    with open(d + "/edge.fq", "w") as f:
        f.write("@e1\nACGT\n+\nIIII\n@e2\n%s\n+\n%s\n@e3\n%s\n+\n%s\n" % ("N" * 60, "I" * 60, "ACGTTGCATG" * 2, "I" * 20))
    return d


def test_sam_valid_and_mostly_correct(data):
    genome = samcheck.load_genome(data + "/g.fa")
    for args, rl in ((["g", "a1.fq", "a2.fq"], 100), (["g", "b1.fq", "b2.fq"], 150), (["-b", "1", "g", "a1.fq", "a2.fq"], 100)):
        hdr, recs = samcheck.parse_sam(run(["-@", "4"] + args, data))
        assert [h for h in hdr if h.startswith("@SQ")] == sorted(h for h in hdr if h.startswith("@SQ"))
        for r in recs:
            samcheck.check_record(r, genome, rl)
        samcheck.check_pairs(recs)
        prim = [r for r in recs if not r["flag"] & 0x900]
        assert len(prim) == 3000
        mapped = [r for r in prim if not r["flag"] & 4]
        assert len(mapped) > 0.97 * len(prim)
        assert np.mean([r["mapq"] >= 40 for r in mapped]) > 0.85
        assert all(r["tags"].get("YD") in ("f", "r", "u") for r in mapped)
    # directional simple pairs: nearly all proper pairs
    hdr, recs = samcheck.parse_sam(run(["-@", "4", "g", "a1.fq", "a2.fq"], data))
    assert np.mean([bool(r["flag"] & 2) for r in recs if not r["flag"] & 0x900]) > 0.95


def test_threads_do_not_change_output(data):
    a = run(["-@", "1", "g", "b1.fq", "b2.fq"], data)
    os.environ["BSX_HOST_THREADS"] = "7"
    try:
        b = run(["-@", "1", "g", "b1.fq", "b2.fq"], data)
    finally:
        del os.environ["BSX_HOST_THREADS"]
    assert a == b


def test_options_take_effect_and_edge_reads(data):
    base = run(["-@", "4", "g", "b1.fq", "b2.fq"], data)
    for extra in (["-a"], ["-b", "1"], ["-S", "-P"], ["-T", "60"], ["-Y"], ["-I", "300,50"], ["-k", "25"]):
        assert run(["-@", "4"] + extra + ["g", "b1.fq", "b2.fq"], data) != base, extra
    assert "\tRG:Z:x" in run(["-@", "2", "-R", "@RG\\tID:x\\tSM:y", "g", "a1.fq"], data)
    hdr, recs = samcheck.parse_sam(run(["-@", "2", "g", "edge.fq"], data))
    assert len(recs) == 3 and all(r["flag"] & 4 for r in recs)
    hdr, recs = samcheck.parse_sam(run(["-@", "2", "-F", "g", "a1.fq"], data))
    assert not hdr
    hdr, recs = samcheck.parse_sam(run(["-@", "4", "g", "long.fq"], data))
    genome = samcheck.load_genome(data + "/g.fa")
    for r in recs:
        samcheck.check_record(r, genome)
    assert sum(1 for r in recs if not r["flag"] & 0x904) >= 38
    one = run(["-1", "ACGTTGCATGACGTTGCATGACGTTGCATGACGTTGCATG", "g"], data)
    assert "inputread" in one


def test_regression_fixture():
    """committed FASTA/FASTQ -> committed SAM (see module docstring for what this fixture is)"""
    import tempfile
    d = tempfile.mkdtemp()
    Index.build(os.path.join(GOLD, "g24k.fa"), d + "/g").close()
    got = run(["-@", "2", d + "/g", os.path.join(GOLD, "pe_small_1.fq"), os.path.join(GOLD, "pe_small_2.fq")], d)
    want = open(os.path.join(GOLD, "pe_small.sam")).read()
    assert got.strip() == want.strip()
