"""-m gpu: the chaining tiers' LDS discipline, on debug builds of the library (tools/dbg/lds_variants.sh -> tests/_build/lds_dbg_<n>/):
  1  guard words before, between and after every LDS object of k_regions / k_regions_mid, checked when a workgroup ends;
  2  the wave's tables filled with 0xff before every strand search (an answer that changes with the fill reads LDS it did not write);
both with the check of every exported chain record ahead of k_c2r (shim.hip, BSX_DEBUG_XCHECK).  Kilobase reads, mixed 300-1024-base
chunks and ordinary pairs, with and without the kernels' cycle counters ($BSX_PHASES): the combination that faulted in round 4 when the
tiers lost their cal_max_gap table -- a miscompiled profiling path, not an LDS overrun (tests/test_isa_exec_join.py looks for
its shape in the assembly).  The SAM must be the product's, byte for byte."""
import os
import subprocess
import pytest
import simdata

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "biscuit_amd", "biscuit_align")


def variant_dir(v):
    d = os.path.join(ROOT, "tests", "_build", "lds_dbg_%d" % v)
    lib = os.path.join(d, "libbiscuit_amd.so")
    src = [os.path.join(ROOT, "biscuit_amd", "csrc", "hip", f) for f in ("k_regions.hip", "shim.hip", "rgx.hpp", "ext_dp.hpp", "dev_common.hpp")]
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in src):
        p = subprocess.run([os.path.join(ROOT, "tools", "dbg", "lds_variants.sh"), str(v)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
    return d


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    from biscuit_amd.api import Index
    d = str(tmp_path_factory.mktemp("ldsdbg"))
    contigs = simdata.make_genome(1000000, seed=21, n_contigs=3)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, 300, 1000, 5))
    mixed = []
    for k, ln in enumerate((300, 420, 600, 760, 900, 1024)):
        mixed += [("m%d_%s" % (ln, n), q) for n, q in simdata.make_single(contigs, 120, ln, 40 + k)]
    simdata.write_fastq(d + "/mixed.fq", mixed)
    p150 = simdata.make_pairs(contigs, 3000, 150, 2, sub=0.01, indel=0.006, pbat_frac=0.3, chimera_frac=0.06, bad_mate_frac=0.06, n_frac=0.03)
    simdata.write_fastq(d + "/b1.fq", [(n, a) for n, a, b in p150])
    simdata.write_fastq(d + "/b2.fq", [(n, b) for n, a, b in p150])
    return d


def run(args, cwd, env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([HIP] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, env=e)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout, p.stderr.decode()


CASES = [("long_1kb", ["-@", "4", "g", "long.fq"]), ("long_mixed_lengths", ["-@", "4", "g", "mixed.fq"]), ("pe150", ["-@", "4", "g", "b1.fq", "b2.fq"]),
         ("pe150_seed_sw_filter", ["-@", "4", "-W", "5", "g", "b1.fq", "b2.fq"])]


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("name,args", CASES, ids=[c[0] for c in CASES])
def test_guards_fill_and_records(data, name, args, variant):
    lib = variant_dir(variant)
    want, _ = run(args, data, {})
    want = b"\n".join(l for l in want.split(b"\n") if not l.startswith(b"@PG"))
    for phases in (None, "1"):
        env = {"LD_LIBRARY_PATH": lib + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")}
        if phases:
            env["BSX_PHASES"] = phases
        out, err = run(args, data, env)
        assert b"LDS GUARD" not in out, out[out.index(b"LDS GUARD"):][:400]
        got = b"\n".join(l for l in out.split(b"\n") if not l.startswith(b"@PG"))
        assert got == want, (name, variant, phases)
        if "long" in name or "filter" in name:   # the sequence in which every tier exports: the records are looked at ahead of k_c2r
            assert "[xcheck]" in err and "[xcheck] 0 bad" in err, err[-1500:]
