"""Implementation-independent validity checks of SAM records against the genome."""
import re
import numpy as np

CIG = re.compile(r"(\d+)([MIDSH])")


def parse_sam(text):
    hdr, recs = [], []
    for l in text.split("\n"):
        if not l:
            continue
        if l.startswith("@"):
            hdr.append(l)
            continue
        f = l.split("\t")
        tags = {t[:2]: t[5:] for t in f[11:]}
        recs.append(dict(qname=f[0], flag=int(f[1]), rname=f[2], pos=int(f[3]), mapq=int(f[4]), cigar=f[5], rnext=f[6], pnext=int(f[7]),
                         tlen=int(f[8]), seq=f[9], qual=f[10], tags=tags, line=l))
    return hdr, recs


def check_record(r, genome, read_len=None):
    """genome: dict name -> str (ACGTN).  Verifies CIGAR/SEQ lengths, MD/NM/ZC against the reference."""
    if r["flag"] & 4:
        assert r["cigar"] == "*"
        return
    ops = [(int(n), o) for n, o in CIG.findall(r["cigar"])]
    assert "".join("%d%s" % x for x in ops) == r["cigar"], r["cigar"]
    qlen = sum(n for n, o in ops if o in "MIS")
    full = sum(n for n, o in ops if o in "MISH")
    if r["seq"] != "*":
        assert len(r["seq"]) == qlen == len(r["qual"]), r["line"]
    if read_len:
        assert full == read_len, (full, read_len, r["line"])
    ref = genome[r["rname"]]
    rlen = sum(n for n, o in ops if o in "MD")
    assert 1 <= r["pos"] and r["pos"] - 1 + rlen <= len(ref), r["line"]
    assert 0 <= r["mapq"] <= 60
    if r["seq"] == "*":
        return
    # walk the alignment: rebuild MD and count mismatches
    x, y = 0, r["pos"] - 1
    nm_gap = nm_mis = conv = 0
    md, run = [], 0
    conv_pair = {("C", "T"), ("G", "A")}
    for k, (n, o) in enumerate(ops):
        if o in "S":
            x += n
        elif o == "H":
            pass
        elif o == "M":
            for i in range(n):
                q, t = r["seq"][x + i], ref[y + i]
                if q == t:
                    run += 1
                else:
                    md.append(str(run)); md.append(t); run = 0
                    if (t, q) in conv_pair:
                        conv += 1
                    else:
                        nm_mis += 1
            x += n; y += n
        elif o == "I":
            x += n; nm_gap += n
        elif o == "D":
            md.append(str(run)); md.append("^" + ref[y:y + n]); run = 0
            y += n; nm_gap += n
    md.append(str(run))
    tags = r["tags"]
    if "N" not in ref[r["pos"] - 1:r["pos"] - 1 + rlen]:
        assert tags["MD"] == "".join(md), (tags["MD"], "".join(md), r["line"])
        zc = int(tags["ZC"])
        # NM counts non-conversion mismatches + gap bases; a record is on one conversion strand, so the other
        # strand's "conversions" are ordinary mismatches: NM + ZC == all mismatches + gaps
        assert int(tags["NM"]) + zc == nm_mis + conv + nm_gap, r["line"]
    assert int(tags["AS"]) <= qlen


def check_pairs(recs):
    by = {}
    for r in recs:
        if r["flag"] & 0x900:
            continue
        by.setdefault(r["qname"], []).append(r)
    for name, rs in by.items():
        if len(rs) != 2:
            continue
        a, b = rs
        assert (a["flag"] & 0x40) and (b["flag"] & 0x80)
        for p, m in ((a, b), (b, a)):
            if not (m["flag"] & 4) and not (p["flag"] & 4):
                assert p["pnext"] == m["pos"], (p["line"], m["line"])
                assert bool(p["flag"] & 0x20) == bool(m["flag"] & 0x10)
        if a["tlen"] or b["tlen"]:
            assert a["tlen"] == b["tlen"]


def load_genome(fa):
    g, name = {}, None
    for l in open(fa):
        l = l.strip()
        if l.startswith(">"):
            name = l[1:].split()[0]; g[name] = []
        elif name:
            g[name].append(l.upper())
    return {k: "".join(v) for k, v in g.items()}
