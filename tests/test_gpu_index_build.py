"""-m gpu: the device index builder (csrc/hip/k_index.hip: 64-bit prefix-doubling suffix sort, BWT blocks, SA samples)
against the host builder (SA-IS, csrc/host/index.c), whose files the reference's own bwt_restore_*/bwt_cal_sa/is_bwt
accept and reproduce (tests/test_oracle_vs_ref.py).  BWT and suffix array of a text are unique: the seven files must be
byte-identical.  Genomes are chosen to reach every branch of the sorter: several batches and slices (small
the setting index_batch), many doubling rounds (long exact repeats, tandem repeats, homopolymers), ties that run into the end
of the text (the sentinel rule), N runs, several contigs."""
import filecmp
import os
import numpy as np
import pytest
import simdata
from biscuit_amd.api import Index, Device

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = [".par.bwt", ".par.sa", ".dau.bwt", ".dau.sa", ".bis.pac", ".bis.ann", ".bis.amb"]


def _both(tmp, fasta, env=None):
    from biscuit_amd import _lib as B_
    old = {}
    for k, v in (env or {}).items():   # settings of the library (csrc/host/tune.c), for the length of the call
        old[k] = None
        B_.tune(k, v)
    try:
        Index.build(fasta, tmp + "/host").close()
        idx = Index.from_fasta(fasta)
        dev = Device(0)
        dev.build_index(idx, fill_host=True)
        idx.save(tmp + "/dev")
        dev.close()
        idx.close()
    finally:
        for k in old:
            B_.tune(k, None)
    for ext in FILES:
        assert filecmp.cmp(tmp + "/host" + ext, tmp + "/dev" + ext, shallow=False), "file %s differs" % ext


def test_golden_fasta(tmp_path):
    _both(str(tmp_path), os.path.join(HERE, "golden", "g24k.fa"))


def test_repeats_and_n_runs_many_batches(tmp_path):
    d = str(tmp_path)
    simdata.write_genome(d + "/g.fa", simdata.make_genome(1500000, seed=77, n_contigs=3))
    _both(d, d + "/g.fa", {"index_batch": "50000"})   # dozens of batches in round 0, slices in the later rounds


def _write(fa, contigs):
    with open(fa, "w") as f:
        for name, s in contigs:
            f.write(">%s\n" % name)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")


def test_deep_repeats_and_sentinel_ties(tmp_path):
    """exact repeats of tens of kb (many doubling rounds), tandem arrays, homopolymers, and a text that ends in a run of
    one base so that tied suffixes run past its end on both strands"""
    rng = np.random.default_rng(5)
    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    unit = rnd(40000)
    c1 = "T" * 300 + rnd(5000) + unit + rnd(3000) + unit + "AC" * 4000 + rnd(100) + "A" * 9000 + rnd(2000) + "ACG" * 3000 + unit[:20000] + "A" * 200
    c2 = "G" * 77 + rnd(1000) + "GATTACA" * 1500 + rnd(500) + "C" * 2100 + "T" * 31
    d = str(tmp_path)
    _write(d + "/g.fa", [("c1", c1), ("c2", c2)])
    _both(d, d + "/g.fa", {"index_batch": "30000"})
    _both(d, d + "/g.fa")


def test_tiny_and_odd_lengths(tmp_path):
    rng = np.random.default_rng(9)
    for n in (64, 65, 127, 128, 129, 1000, 4099):
        d = str(tmp_path / ("n%d" % n))
        os.makedirs(d)
        _write(d + "/g.fa", [("c", "".join("ACGT"[i] for i in rng.integers(0, 4, n)))])
        _both(d, d + "/g.fa")


def test_32mbp_device_vs_host(tmp_path):
    d = str(tmp_path)
    idx = Index.synthetic(32_000_000, seed=11, n_contigs=4)
    idx.build_host()
    idx.save(d + "/host")
    idx.close()
    idx = Index.synthetic(32_000_000, seed=11, n_contigs=4)
    dev = Device(0)
    dev.build_index(idx, fill_host=True)
    idx.save(d + "/dev")
    dev.close(); idx.close()
    for ext in FILES:
        assert filecmp.cmp(d + "/host" + ext, d + "/dev" + ext, shallow=False), "file %s differs" % ext
