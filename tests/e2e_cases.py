"""Shared by tests/test_oracle_e2e.py (CPU suite) and tests/test_gpu_e2e.py (-m gpu): data sets and command lines for the
end-to-end comparison with oracle/e2e.py, the restatement of `biscuit align` that shares no host code with the product."""
import os
import subprocess
import sys
import numpy as np
import simdata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E2E = os.path.join(ROOT, "oracle", "e2e.py")


def strip_pg(b):
    return b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))


def run_exe(exe, args, cwd, env=None, timeout=1800):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=e)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    return strip_pg(p.stdout)


def run_e2e(args, cwd, procs=None):
    """oracle/e2e.py in its own interpreter (it forks workers; the test process may hold a HIP context)"""
    procs = procs or max(1, min(16, len(os.sched_getaffinity(0))))
    return run_exe(sys.executable, [E2E] + args, cwd, env={"E2E_PROCS": str(procs)})


def assert_same_sam(got, want, what):
    if got == want:
        return
    gl, wl = got.split(b"\n"), want.split(b"\n")
    for i, (a, b) in enumerate(zip(gl, wl)):
        assert a == b, "%s: first difference at line %d:\n got: %s\nwant: %s" % (what, i, a[:700], b[:700])
    assert len(gl) == len(wl), "%s: %d lines against %d" % (what, len(gl), len(wl))


def write_fastq_q(path, recs, seed):
    """FASTQ with varied qualities (runs of low quality at either end of some reads) so that -z clips something"""
    r = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for name, seq in recs:
            q = r.integers(30, 41, len(seq))
            if r.random() < 0.4:
                q[:int(r.integers(1, 12))] = r.integers(2, 12)
            if r.random() < 0.4:
                q[-int(r.integers(1, 25)):] = r.integers(2, 12)
            if r.random() < 0.02:
                q[:] = 3
            f.write(b"@" + name.encode() + b"\n" + simdata.BASES[seq].tobytes() + b"\n+\n" + (q + 33).astype(np.uint8).tobytes() + b"\n")


def make_data(d, genome_bp, n_pairs, n_long, repeat_frac=0.05, seed=21, n_contigs=3):
    """genome + index files (the repository's builder: the files are the pinned part, tests/test_oracle_vs_ref.py) + read sets:
    b1/b2 2x150 hard pairs, a1/a2 2x100 clean pairs, q1/q2 = b with varied qualities and adaptor tails, long.fq 1 kb reads,
    bc1/bc2 barcoded names with comments, il.fq interleaved pairs and singletons, edge.fq degenerate reads"""
    from biscuit_amd.api import Index
    contigs = simdata.make_genome(genome_bp, seed=seed, n_contigs=n_contigs, repeat_frac=repeat_frac)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    hard = simdata.make_pairs(contigs, n_pairs, 150, seed + 1, sub=0.01, indel=0.006, pbat_frac=0.3, chimera_frac=0.06, bad_mate_frac=0.06, n_frac=0.03)
    clean = simdata.make_pairs(contigs, n_pairs, 100, seed + 2, frag=(180, 320), sub=0.005)
    for tag, ps in (("a", clean), ("b", hard)):
        simdata.write_fastq(d + "/%s1.fq" % tag, [(n, a) for n, a, b in ps])
        simdata.write_fastq(d + "/%s2.fq" % tag, [(n, b) for n, a, b in ps])
    ad = np.array([0, 2, 0, 3, 1, 2, 2, 0, 0, 2, 0, 2, 1], np.uint8)      # AGATCGGAAGAGC
    r = np.random.default_rng(seed + 3)
    qrecs = [[], []]
    for n, a, b in hard[:max(200, n_pairs // 2)]:
        for k, s in enumerate((a, b)):
            if r.random() < 0.3:      # adaptor read-through: the tail of the read is (part of) the adaptor
                cut = int(r.integers(len(s) - 40, len(s) - 3))
                s = np.concatenate([s[:cut], ad])[:len(s)]
            qrecs[k].append((n, s))
    write_fastq_q(d + "/q1.fq", qrecs[0], seed + 4)
    write_fastq_q(d + "/q2.fq", qrecs[1], seed + 5)
    simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, n_long, 1000, seed + 6))
    nb = max(100, n_pairs // 8)
    with open(d + "/bc1.fq", "wb") as f1, open(d + "/bc2.fq", "wb") as f2:
        for i, (n, a, b) in enumerate(hard[:nb]):
            nm = ("%s_ACGTACGT_TTGCA%d" % (n, i % 7)).encode()
            for f, s, k in ((f1, a, 1), (f2, b, 2)):
                f.write(b"@" + nm + b"/%d extra:comment %d\n" % (k, i) + simdata.BASES[s].tobytes() + b"\n+\n" + b"F" * len(s) + b"\n")
    with open(d + "/il.fq", "wb") as f:      # -p: pairs interleaved with singletons
        for i, (n, a, b) in enumerate(hard[:nb]):
            recs = [(n, a)] if i % 5 == 3 else [(n, a), (n, b)]
            for nm, s in recs:
                f.write(b"@" + nm.encode() + b"\n" + simdata.BASES[s].tobytes() + b"\n+\n" + b"I" * len(s) + b"\n")
    g0 = contigs[0][1]
    with open(d + "/edge.fq", "w") as f:
        f.write("@e1\nACGT\n+\nIIII\n@e2\n%s\n+\n%s\n@e3\n%s\n+\n%s\n" % ("N" * 60, "I" * 60, "ACGTTGCATG" * 2, "I" * 20))
        s = simdata.BASES[g0[5000:5150]].tobytes().decode()
        f.write("@e4_exact\n%s\n+\n%s\n" % (s, "I" * 150))
        f.write("@e5_lowq\n%s\n+\n%s\n" % (s, "#" * 150))
        f.write(">e6_fasta_record\n%s\n" % s[:90])
    with open(d + "/g.alt", "w") as f:      # bns_restore reads <prefix>.alt (bntseq.c:189-214); moved aside unless a case asks for it
        f.write("@comment line\n%s\textra column\n" % contigs[-1][0])
    os.rename(d + "/g.alt", d + "/g.alt.off")
    return contigs


# name, command line, needs the .alt file
PE = ["b1.fq", "b2.fq"]
CASES_CORE = [
    ("pe150_b0", ["-@", "4", "g"] + PE),
    ("pe150_b1", ["-@", "4", "-b", "1", "g"] + PE),
    ("se150", ["-@", "4", "g", "b1.fq"]),
    ("pe150_all_softclip", ["-@", "4", "-a", "-Y", "g"] + PE),
    ("long_1kb", ["-@", "4", "g", "long.fq"]),
]
CASES_MORE = [
    ("pe100_default", ["-@", "4", "g", "a1.fq", "a2.fq"]),
    ("pe150_norescue_nopair", ["-@", "4", "-S", "-P", "g"] + PE),
    ("pe150_fixed_isize_rg", ["-@", "4", "-I", "350,60", "-R", "@RG\\tID:x\\tSM:y", "-C", "g"] + PE),
    ("se150_daughter", ["-@", "4", "-b", "3", "g", "b2.fq"]),
    ("pe150_clip_qual_adaptor", ["-@", "4", "-J", "AGATCGGAAGAGC", "-K", "AGATCGGAAGAGC", "-z", "15", "-5", "2", "-3", "1", "g", "q1.fq", "q2.fq"]),
    ("se150_scoring", ["-@", "4", "-A", "2", "-B", "3", "-O", "5,7", "-E", "2,1", "-L", "4,6", "-T", "40", "-k", "17", "-w", "60", "g", "b1.fq"]),
    ("pe150_chain_knobs", ["-@", "4", "-c", "12", "-D", "0.3", "-W", "25", "-m", "30", "-G", "4000", "g"] + PE),
    ("se150_bsstrand_band", ["-@", "4", "-f", "1", "-w", "12", "-L", "0,9", "-r", "1.2", "-y", "30", "g", "b1.fq"]),
    ("pe150_barcode_comment_hdr", ["-@", "4", "-9", "-C", "-V", "-H", "@CO\\tfrom a test", "g", "bc1.fq", "bc2.fq"]),
    ("smart_pairing", ["-@", "4", "-p", "g", "il.fq"]),
    ("edge_reads", ["-@", "4", "g", "edge.fq"]),
    ("pe150_supp_mapq_split", ["-@", "4", "-q", "-U", "9", "-s", "3", "-N", "4", "-X", "0.4", "-g", "2,1", "-Q", "0", "g"] + PE),
    ("pe150_two_chunks", ["-@", "1", "g"] + PE),      # chunks of 10 Mbp: only splits when the set is larger than that
]
CASES_ALT = [
    ("pe150_alt_contig", ["-@", "4", "g"] + PE),
    ("se150_alt_ignored", ["-@", "4", "-j", "g", "b1.fq"]),
]
