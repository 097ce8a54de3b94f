"""Seeded synthetic genome / bisulfite read generator used by tests and bench (numpy only).

Genomes are i.i.d. bases plus planted repeat families, tandem repeats and an N run (a purely random
genome never exercises secondary hits, mate rescue or region merging: SURVEY appendix C).  Reads
follow the directional protocol: R1 = bisulfite-converted strand, R2 = reverse complement of the
fragment; C->T conversion with retention 0.70 at CpG and 0.01 elsewhere.
"""
import numpy as np

BASES = np.frombuffer(b"ACGTN", dtype=np.uint8)


def make_genome(n, seed, n_contigs=2, repeat_frac=0.05, n_run=True):
    r = np.random.default_rng(seed)
    g = r.integers(0, 4, n).astype(np.uint8)
    planted = 0
    while planted < n * repeat_frac:
        l = int(r.integers(200, 2000))
        s = int(r.integers(0, n - l))
        for _ in range(int(r.integers(1, 5))):
            d = int(r.integers(0, n - l))
            copy = g[s:s + l].copy()
            div = r.random(l) < r.uniform(0.0, 0.05)
            copy[div] = (copy[div] + r.integers(1, 4, int(div.sum()))) % 4
            if r.random() < 0.3:
                copy = (3 - copy[::-1]).astype(np.uint8)
            g[d:d + l] = copy
            planted += l
    for _ in range(max(1, n // 100000)):  # tandem repeats
        unit = r.integers(0, 4, int(r.integers(2, 30))).astype(np.uint8)
        reps = int(r.integers(5, 40))
        t = np.tile(unit, reps)
        d = int(r.integers(0, n - len(t)))
        g[d:d + len(t)] = t
    if n_run and n > 20000:
        d = int(r.integers(n // 4, n // 2))
        g[d:d + min(1000, n // 100)] = 4
    cuts = [0] + sorted(int(x) for x in r.choice(np.arange(n // 10, n - n // 10), n_contigs - 1, replace=False)) + [n]
    return [("chr%d" % (i + 1), g[cuts[i]:cuts[i + 1]]) for i in range(n_contigs)]


def write_genome(path, contigs):
    with open(path, "wb") as f:
        for name, g in contigs:
            f.write(b">" + name.encode() + b"\n")
            s = BASES[g].tobytes()
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + b"\n")


def revcomp(x):
    y = x[::-1].copy()
    m = y < 4
    y[m] = 3 - y[m]
    return y


def bisulfite(frag, r, cpg_ret=0.70, other_ret=0.01):
    out = frag.copy()
    isc = frag == 1
    nxt = np.append(frag[1:], 4)
    cpg = isc & (nxt == 2)
    keep = np.where(cpg, r.random(len(frag)) < cpg_ret, r.random(len(frag)) < other_ret)
    out[isc & ~keep] = 3
    return out


def mutate(seq, r, sub, indel):
    out = []
    i = 0
    n = len(seq)
    while i < n:
        x = r.random()
        if x < sub:
            out.append((int(seq[i]) + int(r.integers(1, 4))) % 4 if seq[i] < 4 else 4)
            i += 1
        elif x < sub + indel / 2:
            i += 1 + int(r.integers(0, 3))
        elif x < sub + indel:
            out.extend(r.integers(0, 4, int(r.integers(1, 4))).tolist())
        else:
            out.append(int(seq[i]))
            i += 1
    return np.array(out, dtype=np.uint8)


def make_pairs(contigs, n_pairs, read_len, seed, frag=(200, 500), sub=0.005, indel=0.0, pbat_frac=0.0,
               chimera_frac=0.0, bad_mate_frac=0.0, n_frac=0.0):
    """Returns list of (name, r1, r2) with nt4 uint8 arrays."""
    r = np.random.default_rng(seed)
    lens = np.array([len(g) for _, g in contigs])
    out = []
    while len(out) < n_pairs:
        ci = int(r.choice(len(contigs), p=lens / lens.sum()))
        g = contigs[ci][1]
        fl = int(r.integers(frag[0], frag[1] + 1))
        if fl >= len(g):
            continue
        s = int(r.integers(0, len(g) - fl))
        f = g[s:s + fl].copy()
        if (f == 4).mean() > 0.2:
            continue
        if r.random() < chimera_frac:  # chimeric fragment: second half from elsewhere
            cj = int(r.integers(0, len(contigs)))
            g2 = contigs[cj][1]
            s2 = int(r.integers(0, len(g2) - fl))
            f[fl // 2:] = g2[s2:s2 + fl - fl // 2]
        if r.random() < 0.5:
            f = revcomp(f)
        conv = bisulfite(f, r)
        if sub or indel:
            conv = mutate(conv, r, sub, indel)
        if len(conv) < read_len:
            continue
        r1 = conv[:read_len].copy()
        r2 = revcomp(conv)[:read_len].copy()
        if r.random() < pbat_frac:
            r1, r2 = r2, r1
        if r.random() < bad_mate_frac:  # degrade one mate
            k = r.random(read_len) < 0.15
            r2[k] = (r2[k] + r.integers(1, 4, int(k.sum()))) % 4
        if n_frac and r.random() < n_frac:
            r1[int(r.integers(0, read_len))] = 4
        out.append(("r%06d" % len(out), r1, r2))
    return out


def make_single(contigs, n, read_len, seed, sub=0.01, indel=0.004):
    r = np.random.default_rng(seed)
    lens = np.array([len(g) for _, g in contigs])
    out = []
    while len(out) < n:
        ci = int(r.choice(len(contigs), p=lens / lens.sum()))
        g = contigs[ci][1]
        fl = read_len + 60
        if fl >= len(g):
            continue
        s = int(r.integers(0, len(g) - fl))
        f = g[s:s + fl].copy()
        if (f == 4).mean() > 0.2:
            continue
        if r.random() < 0.5:
            f = revcomp(f)
        conv = mutate(bisulfite(f, r), r, sub, indel)
        if len(conv) < read_len:
            continue
        out.append(("s%06d" % len(out), conv[:read_len].copy()))
    return out


def write_fastq(path, recs, qual_char=b"I"):
    with open(path, "wb") as f:
        for name, seq in recs:
            f.write(b"@" + name.encode() + b"\n" + BASES[seq].tobytes() + b"\n+\n" + qual_char * len(seq) + b"\n")


def read_buffer(seqs):
    """Concatenate reads into one chunk buffer; returns (buf, offsets)."""
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    for i, s in enumerate(seqs):
        offs[i + 1] = offs[i] + len(s)
    buf = np.concatenate(seqs).astype(np.uint8) if seqs else np.zeros(0, np.uint8)
    return buf, offs
