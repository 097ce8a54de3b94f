"""CPU: oracle/e2e.py -- the end-to-end restatement of `biscuit align` over the reference's own kernels (oracle/_ref), which
shares no host code with the product -- against `oracle_align` (the product's host pipeline over the CPU restatement of the device
kernels, oracle/port.c).  Two implementations of everything between FASTQ and SAM written from the same reference lines must give
the same bytes.  The -m gpu counterpart (tests/test_gpu_e2e.py) puts the HIP path on the other side."""
import os
import pytest
import e2e_cases as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "oracle", "oracle_align")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libbiscuit_ref.so")),
                                reason="oracle/_ref (the reference's kernels, built where /root/reference exists) is absent")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("e2e"))
    E.make_data(d, 300000, 700, 60, repeat_frac=0.2)
    return d


@pytest.mark.parametrize("name,args", E.CASES_CORE + E.CASES_MORE, ids=[c[0] for c in E.CASES_CORE + E.CASES_MORE])
def test_e2e_equals_host_pipeline(data, name, args):
    want = E.run_e2e(args, data, procs=4)
    got = E.run_exe(CPU, args, data)
    assert got.count(b"\n") >= 6
    E.assert_same_sam(got, want, name)


@pytest.mark.parametrize("name,args", E.CASES_ALT, ids=[c[0] for c in E.CASES_ALT])
def test_e2e_alt_contigs(data, name, args):
    os.rename(data + "/g.alt.off", data + "/g.alt")
    try:
        want = E.run_e2e(args, data, procs=4)
        got = E.run_exe(CPU, args, data)
    finally:
        os.rename(data + "/g.alt", data + "/g.alt.off")
    E.assert_same_sam(got, want, name)


@pytest.mark.parametrize("name,args", E.CASES_CORE + E.CASES_MORE[4:6], ids=[c[0] for c in E.CASES_CORE + E.CASES_MORE[4:6]])
def test_host_pipeline_over_reference_kernels(data, name, args):
    """bench.py's cpu_baseline: the host pipeline over the reference's OWN kernels (ORACLE_REF_KERNELS: oracle/_ref's bwt_smem1a,
    bwt_seed_strategy1, bwt_sa, ksw_extend2, ksw_align2 (SSE2), ksw_global2 instead of oracle/port.c's scalar restatements) gives the
    same SAM as the end-to-end oracle."""
    want = E.run_e2e(args, data, procs=4)
    got = E.run_exe(CPU, args, data, env={"ORACLE_REF_KERNELS": "1"})
    E.assert_same_sam(got, want, name)
