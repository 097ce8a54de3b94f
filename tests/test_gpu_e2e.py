"""-m gpu: the product (`biscuit_align`: HIP kernels + the C host pipeline) against oracle/e2e.py, the end-to-end restatement of
`biscuit align` over the reference's own kernels (oracle/_ref) that shares NO host code with the product: chaining, chain filter,
seed filter, extension bookkeeping, de-duplication and merging, insert-size statistics, mate rescue, primary marking, pairing,
MAPQ, CIGAR/MD/NM/ZC/ZR, SA/XA/XB and the SAM text are all computed twice, independently, and must agree byte for byte.
At least 2 000 pairs (or reads) per mode: paired-end -b 0 / -b 1, single-end, -a -Y, 1 kb reads, on a genome with few repeats and
on a repeat-rich one (35 % planted repeats: a dozen chains per strand search, XA/SA tags on a third of the records)."""
import os
import pytest
import e2e_cases as E

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "biscuit_amd", "biscuit_align")


@pytest.fixture(scope="module")
def std(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("e2e_std"))
    E.make_data(d, 1000000, 2500, 2000, repeat_frac=0.05)
    return d


@pytest.fixture(scope="module")
def rich(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("e2e_rich"))
    E.make_data(d, 400000, 6000, 2000, repeat_frac=0.35, seed=5, n_contigs=5)
    return d


def _compare(d, name, args):
    want = E.run_e2e(args, d)
    got = E.run_exe(HIP, args, d)
    assert got.count(b"\n") >= 6
    E.assert_same_sam(got, want, name)
    return got


ALL = E.CASES_CORE + E.CASES_MORE


@pytest.mark.parametrize("name,args", ALL, ids=[c[0] for c in ALL])
def test_hip_equals_e2e_oracle(std, name, args):
    _compare(std, name, args)


@pytest.mark.parametrize("name,args", E.CASES_CORE + E.CASES_MORE[1:3] + E.CASES_MORE[6:7], ids=[c[0] for c in E.CASES_CORE + E.CASES_MORE[1:3] + E.CASES_MORE[6:7]])
def test_hip_equals_e2e_oracle_repeat_rich(rich, name, args):
    got = _compare(rich, name, args)
    if name == "pe150_b0":      # the data does what it is for
        assert got.count(b"XA:Z:") > 2000 and got.count(b"SA:Z:") > 100


@pytest.mark.parametrize("name,args", E.CASES_ALT, ids=[c[0] for c in E.CASES_ALT])
def test_hip_equals_e2e_oracle_alt_contigs(rich, name, args):
    os.rename(rich + "/g.alt.off", rich + "/g.alt")
    try:
        got = _compare(rich, name, args)
    finally:
        os.rename(rich + "/g.alt", rich + "/g.alt.off")
    if name == "pe150_alt_contig":
        assert got.count(b"PA:f:") > 50


@pytest.fixture(scope="module")
def hard(tmp_path_factory):
    """a genome with the high-copy interspersed repeat families of a mammalian one (csrc/host/sim.c profile 1: SINE-, LINE-, LTR-like
    families and satellite arrays, ~43 % repeats; 8 Mbp, so the SINE-like family has 2 700 copies) and pairs simulated from it"""
    import ctypes as C
    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index
    d = str(tmp_path_factory.mktemp("e2e_hard"))
    L = B.lib()
    L.bsx_sim_genome2.argtypes = [C.c_char_p, C.c_int64, C.c_uint64, C.c_int, C.c_double, C.c_int]
    B.check(L.bsx_sim_genome2((d + "/g.fa").encode(), 8000000, 77, 5, 0.05, 1), "sim_genome2")
    B.check(L.bsx_index_build((d + "/g.fa").encode(), (d + "/g").encode()), "index_build")
    idx = Index(d + "/g")
    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_sim_write_fastq.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_int]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, 4000, 150, 5, 200, 500, 0.005, 0.1, C.byref(p)), "sim_pairs")
    B.check(L.bsx_sim_write_fastq(p, 8000, (d + "/b1.fq").encode(), (d + "/b2.fq").encode(), 0), "write_fastq")
    L.bsx_sim_free_reads(p, 8000)
    idx.close()
    return d


HARD_CASES = [
    ("pe150_b0", ["-@", "4", "g", "b1.fq", "b2.fq"]),
    ("pe150_b1", ["-@", "4", "-b", "1", "g", "b1.fq", "b2.fq"]),
    ("se150_all_softclip", ["-@", "4", "-a", "-Y", "g", "b1.fq"]),
    ("pe150_max_occ_100", ["-@", "4", "-c", "100", "g", "b1.fq", "b2.fq"]),      # SA intervals beyond max_occ: the rule of memchain.c:325-326
    ("pe150_max_occ_20", ["-@", "4", "-c", "20", "-y", "8", "g", "b1.fq", "b2.fq"]),
]


@pytest.mark.parametrize("name,args", HARD_CASES, ids=[c[0] for c in HARD_CASES])
def test_hip_equals_e2e_oracle_high_copy_repeats(hard, name, args):
    got = _compare(hard, name, args)
    if name == "pe150_b0":
        assert got.count(b"XA:Z:") + got.count(b"XB:Z:") > 300
