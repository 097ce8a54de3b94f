import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_index(tmp_path_factory):
    """1 contig-pair, 120 kb genome with planted repeats + an N run; index built by the repo's own builder."""
    import simdata
    from biscuit_amd.api import Index
    d = tmp_path_factory.mktemp("idx")
    fa = str(d / "g.fa")
    simdata.write_genome(fa, simdata.make_genome(120000, seed=21, n_contigs=2))
    return Index.build(fa, str(d / "g"))


@pytest.fixture(scope="session")
def port(small_index):
    import oracle_lib
    return oracle_lib.Port(small_index, n_threads=4)


@pytest.fixture(scope="session")
def device(small_index):
    from biscuit_amd.api import Device
    dev = Device(0)   # raises loudly when there is no HIP device
    dev.upload_index(small_index)
    return dev


@pytest.fixture
def tune():
    """set settings of the library (biscuit_amd/csrc/host/tune.c) for the length of a test: tune(name, value)"""
    from biscuit_amd import _lib as B
    names = []

    def set_(name, value):
        names.append(name)
        B.tune(name, value)
    yield set_
    for n in names:
        B.tune(n, None)
