"""CPU: the command line's I/O error behaviour (csrc/host/cli.c is shared by the product entry and the checker's entry):
a failed write to stdout must end in a non-zero exit status, as the reference's err_fputs/err_fflush do (utils.c:214-240)."""
import os
import subprocess
import simdata
from biscuit_amd.api import Index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_write_error_is_reported(tmp_path):
    d = str(tmp_path)
    contigs = simdata.make_genome(60000, seed=8, n_contigs=1)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    pairs = simdata.make_pairs(contigs, 400, 100, 9, frag=(150, 300))
    simdata.write_fastq(d + "/r1.fq", [(n, a) for n, a, b in pairs])
    exe = os.path.join(ROOT, "oracle", "oracle_align")
    ok = subprocess.run([exe, "-@", "1", "g", "r1.fq"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert ok.returncode == 0 and ok.stdout.count(b"\n") > 400
    with open("/dev/full", "wb") as full:
        bad = subprocess.run([exe, "-@", "1", "g", "r1.fq"], cwd=d, stdout=full, stderr=subprocess.PIPE, timeout=300)
    assert bad.returncode != 0
    assert b"failed to write" in bad.stderr
