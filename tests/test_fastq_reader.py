"""CPU: the FASTA/FASTQ reader (csrc/host/fastq.c) against a direct Python statement of klib's kseq_read grammar
(lib/aln/kseq.h:182-222) and of bis_bseq_read's chunk rule (lib/aln/bwa.c:817-850)."""
import ctypes as C
import gzip
import random
from biscuit_amd import _lib as B


def kseq_records(text):
    """kseq_read: header '>' or '@', name to the first white space, comment = rest of line, sequence lines until a line
    starting with '>', '@' or '+', then quality lines until as many characters as bases."""
    recs, i, n = [], 0, len(text)
    while True:
        while i < n and text[i] not in ">@":
            i += 1
        if i >= n:
            break
        i += 1
        j = i
        while j < n and not text[j].isspace():
            j += 1
        name = text[i:j]
        comment = ""
        if j < n and text[j] != "\n":
            k = text.find("\n", j + 1)
            k = n if k < 0 else k
            comment = text[j + 1:k].rstrip("\r")
            j = k
        i = j + 1
        seq = []
        while i < n and text[i] not in ">+@":
            k = text.find("\n", i)
            k = n if k < 0 else k
            seq.append(text[i:k].rstrip("\r"))
            i = k + 1
        seq = "".join(seq)
        qual = None
        if i < n and text[i] == "+":
            k = text.find("\n", i)
            i = (n if k < 0 else k) + 1
            q = []
            while sum(map(len, q)) < len(seq) and i < n:
                k = text.find("\n", i)
                k = n if k < 0 else k
                q.append(text[i:k].rstrip("\r"))
                i = k + 1
            qual = "".join(q)
            if len(qual) != len(seq):
                break   # truncated quality: the reader stops here
        recs.append((name, comment, seq, qual))
    return recs


def read_all(path1, path2, chunk_size):
    L = B.lib()
    L.bsx_hook_fq_open.restype = C.c_void_p
    L.bsx_hook_fq_open.argtypes = [C.c_char_p]
    L.bsx_hook_fq_close.argtypes = [C.c_void_p]
    L.bsx_hook_fq_chunk.restype = C.POINTER(B.Read)
    L.bsx_hook_fq_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    f1 = L.bsx_hook_fq_open(path1.encode())
    f2 = L.bsx_hook_fq_open(path2.encode()) if path2 else None
    chunks = []
    while True:
        n = C.c_int()
        r = L.bsx_hook_fq_chunk(f1, f2, chunk_size, 0, C.byref(n))
        if not r or n.value == 0:
            break
        out = []
        for i in range(n.value):
            x = r[i]
            out.append((x.name.decode(), (x.comment or b"").decode(), "".join("ACGTN"[x.seq[k]] for k in range(x.l_seq)), x.qual.decode() if x.qual else None))
        chunks.append(out)
        L.bsx_sim_free_reads(r, n.value)
    L.bsx_hook_fq_close(f1)
    if f2:
        L.bsx_hook_fq_close(f2)
    return chunks


def make_text(rng, n, fastq=True, crlf=False, multiline=False):
    eol = "\r\n" if crlf else "\n"
    out = []
    for i in range(n):
        l = rng.choice([0, 1, 7, 60, 61, 150, 300]) if i % 11 == 0 else rng.randint(20, 200)
        seq = "".join(rng.choice("ACGTN") for _ in range(l))
        name = "r%d" % i + ("/1" if i % 3 == 0 else "")
        com = " some comment %d" % i if i % 4 == 0 else ""
        width = 60 if multiline else max(1, l)
        lines = [seq[k:k + width] for k in range(0, l, width)] or [""]
        if fastq:
            qual = "".join(rng.choice("!#5I@>+") for _ in range(l))   # '@', '>' and '+' may start a quality line
            qlines = [qual[k:k + width] for k in range(0, l, width)] or [""]
            out.append("@" + name + com + eol + eol.join(lines) + eol + "+" + (name if i % 5 == 0 else "") + eol + eol.join(qlines) + eol)
        else:
            out.append(">" + name + com + eol + eol.join(lines) + eol)
    return "".join(out)


def check(tmp_path, text, gz=False, chunk_size=2000):
    p = str(tmp_path / ("x.fq.gz" if gz else "x.fq"))
    (gzip.open(p, "wt", newline="") if gz else open(p, "w", newline="")).write(text)
    want = [(n[:-2] if len(n) > 2 and n[-2] == "/" and n[-1].isdigit() else n, c, s.upper().translate(str.maketrans("BDEFHIJKLMOPQRSUVWXYZ", "N" * 21)), q or None)
            for n, c, s, q in kseq_records(text)]
    chunks = read_all(p, None, chunk_size)
    got = [r for ch in chunks for r in ch]
    assert got == want
    # chunk rule: a chunk ends at the first even record count once it holds >= chunk_size bases
    for ch in chunks[:-1]:
        tot = 0
        for k, r in enumerate(ch):
            tot += len(r[2])
            if tot >= chunk_size and (k + 1) % 2 == 0:
                assert k + 1 == len(ch)
                break
        else:
            raise AssertionError("chunk ended early")


def test_fastq_grammar_variants(tmp_path):
    rng = random.Random(5)
    check(tmp_path, make_text(rng, 400))
    check(tmp_path, make_text(rng, 300, crlf=True))
    check(tmp_path, make_text(rng, 300, multiline=True))
    check(tmp_path, make_text(rng, 300, fastq=False, multiline=True))
    check(tmp_path, make_text(rng, 300, multiline=True), gz=True)
    check(tmp_path, make_text(rng, 50).rstrip("\n"))                       # no newline at the end of the file
    check(tmp_path, "garbage before\n" + make_text(rng, 20))
    check(tmp_path, make_text(rng, 5000), chunk_size=100000)               # records straddling the 256 KB buffer


def test_paired_files_interleave(tmp_path):
    rng = random.Random(6)
    a, b = make_text(rng, 200), make_text(rng, 200)
    p1, p2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(p1, "w").write(a)
    open(p2, "w").write(b)
    got = [r for ch in read_all(p1, p2, 3000) for r in ch]
    ra, rb = kseq_records(a), kseq_records(b)
    assert len(got) == 400
    assert [g[2] for g in got[0::2]] == [r[2] for r in ra] and [g[2] for g in got[1::2]] == [r[2] for r in rb]


def read_all_threaded(path1, path2, chunk_size):
    L = B.lib()
    L.bsx_hook_fq_open.restype = C.c_void_p
    L.bsx_hook_fq_open.argtypes = [C.c_char_p]
    L.bsx_hook_fq_close.argtypes = [C.c_void_p]
    L.bsx_hook_fq_pair_open.restype = C.c_void_p
    L.bsx_hook_fq_pair_open.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bsx_hook_fq_pair_close.argtypes = [C.c_void_p]
    L.bsx_hook_fq_pair_chunk.restype = C.POINTER(B.Read)
    L.bsx_hook_fq_pair_chunk.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    f1 = L.bsx_hook_fq_open(path1.encode())
    f2 = L.bsx_hook_fq_open(path2.encode()) if path2 else None
    p = L.bsx_hook_fq_pair_open(f1, f2, 0)
    chunks = []
    while True:
        n = C.c_int()
        r = L.bsx_hook_fq_pair_chunk(p, chunk_size, C.byref(n))
        if not r or n.value == 0:
            break
        chunks.append([(r[i].name.decode(), (r[i].comment or b"").decode(), "".join("ACGTN"[r[i].seq[k]] for k in range(r[i].l_seq)),
                        r[i].qual.decode() if r[i].qual else None, r[i].id) for i in range(n.value)])
        L.bsx_sim_free_reads(r, n.value)
    L.bsx_hook_fq_pair_close(p)
    L.bsx_hook_fq_close(f1)
    if f2:
        L.bsx_hook_fq_close(f2)
    return chunks


def test_parser_threads_give_the_same_chunks(tmp_path):
    """One parser thread per file (what the command line uses) against the in-line reader: same chunks, same records, same ids;
    also with files of unequal length and with the consumer stopping early."""
    rng = random.Random(7)
    a, b = make_text(rng, 9000), make_text(rng, 9000, multiline=True)
    p1, p2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq.gz")
    open(p1, "w").write(a)
    gzip.open(p2, "wt").write(b)
    for chunk in (3000, 50000, 10 ** 7):
        want = [[r + (i,) for i, r in enumerate(ch)] for ch in read_all(p1, p2, chunk)]
        assert read_all_threaded(p1, p2, chunk) == want
        assert read_all_threaded(p1, None, chunk) == [[r + (i,) for i, r in enumerate(ch)] for ch in read_all(p1, None, chunk)]
    short = str(tmp_path / "short.fq")
    open(short, "w").write(make_text(random.Random(8), 100))
    assert sum(len(c) for c in read_all_threaded(p1, short, 4000)) == sum(len(c) for c in read_all(p1, short, 4000)) == 200
    # early close with blocks still queued must not hang or leak into the next reader
    L = B.lib()
    f1 = L.bsx_hook_fq_open(p1.encode())
    p = L.bsx_hook_fq_pair_open(f1, None, 0)
    n = C.c_int()
    r = L.bsx_hook_fq_pair_chunk(p, 100, C.byref(n))
    L.bsx_sim_free_reads(r, n.value)
    L.bsx_hook_fq_pair_close(p)
    L.bsx_hook_fq_close(f1)


class ChunkPos(C.Structure):
    _fields_ = [("off1", C.c_int64), ("off2", C.c_int64), ("n_before", C.c_int64), ("n", C.c_int), ("pad", C.c_int)]


def scan_table(path1, path2, chunk_size):
    L = B.lib()
    L.bsx_fq_scan_table.restype = C.c_int64
    L.bsx_fq_scan_table.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int64, C.POINTER(ChunkPos)]
    tab = (ChunkPos * 4096)()
    k = L.bsx_fq_scan_table(path1.encode(), path2.encode() if path2 else None, chunk_size, 4096, tab)
    return None if k < 0 else [tab[i] for i in range(k)]


def read_chunk_at(path1, path2, off1, off2, chunk_size):
    L = B.lib()
    L.bsx_hook_fq_seek.argtypes = [C.c_void_p, C.c_int64]
    f1 = L.bsx_hook_fq_open(path1.encode())
    f2 = L.bsx_hook_fq_open(path2.encode()) if path2 else None
    assert L.bsx_hook_fq_seek(f1, off1) == 0 and (f2 is None or L.bsx_hook_fq_seek(f2, off2) == 0)
    n = C.c_int()
    r = L.bsx_hook_fq_chunk(f1, f2, chunk_size, 0, C.byref(n))
    out = [(r[i].name.decode(), "".join("ACGTN"[r[i].seq[k]] for k in range(r[i].l_seq)), r[i].qual.decode() if r[i].qual else None) for i in range(n.value)]
    if r:
        L.bsx_sim_free_reads(r, n.value)
    L.bsx_hook_fq_close(f1)
    if f2:
        L.bsx_hook_fq_close(f2)
    return out


def test_chunk_scan_equals_the_reader(tmp_path):
    """Multi-GPU input: the chunk boundaries found by the light scan (no records built: fastq.c, fq_scan) are the chunks the reader
    makes, for every grammar variant; a reader that seeks to a chunk's offsets reads exactly that chunk."""
    rng = random.Random(9)
    texts = [make_text(rng, 400), make_text(rng, 300, crlf=True), make_text(rng, 300, multiline=True), make_text(rng, 300, fastq=False, multiline=True),
             make_text(rng, 50).rstrip("\n"), "garbage before\n" + make_text(rng, 20), make_text(rng, 5000),
             make_text(rng, 40, fastq=False) + make_text(rng, 40) + make_text(rng, 40, fastq=False, multiline=True)]
    for ti, text in enumerate(texts):
        p = str(tmp_path / ("s%d.fq" % ti))
        open(p, "w", newline="").write(text)
        for chunk_size in (700, 2000, 100000):
            chunks = read_all(p, None, chunk_size)
            tab = scan_table(p, None, chunk_size)
            assert [t.n for t in tab] == [len(c) for c in chunks], (ti, chunk_size)
            assert [t.n_before for t in tab] == [sum(len(c) for c in chunks[:k]) for k in range(len(chunks))]
            for k in sorted(set([0, len(tab) // 2, len(tab) - 1])):
                got = read_chunk_at(p, None, tab[k].off1, 0, chunk_size)
                assert got == [(r[0], r[2], r[3]) for r in chunks[k]], (ti, chunk_size, k)
    # two files, records of different byte lengths in each; and one file shorter than the other
    a, b = make_text(rng, 300, multiline=True), make_text(rng, 300, crlf=True)
    p1, p2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(p1, "w", newline="").write(a)
    open(p2, "w", newline="").write(b)
    for chunk_size in (1500, 9000):
        chunks = read_all(p1, p2, chunk_size)
        tab = scan_table(p1, p2, chunk_size)
        assert [t.n for t in tab] == [len(c) for c in chunks]
        for k in range(len(tab)):
            got = read_chunk_at(p1, p2, tab[k].off1, tab[k].off2, chunk_size)
            assert got == [(r[0], r[2], r[3]) for r in chunks[k]], (chunk_size, k)
    open(p2, "w", newline="").write(make_text(random.Random(1), 120))
    chunks = read_all(p1, p2, 1500)
    assert [t.n for t in scan_table(p1, p2, 1500)] == [len(c) for c in chunks] and sum(len(c) for c in chunks) == 240
    # compressed input has no offsets to seek to: no table, the ranks fall back to parsing everything
    pz = str(tmp_path / "z.fq.gz")
    gzip.open(pz, "wt").write(a)
    assert scan_table(pz, None, 2000) is None


def write_bgzf(path, data, block=20000):
    """a BGZF file (what bgzip writes): gzip members of at most 64 KB, each with a 'BC' extra field holding its own size, then the empty
    end-of-file member"""
    import struct
    import zlib
    with open(path, "wb") as f:
        for i in list(range(0, len(data), block)) + [None]:
            raw = b"" if i is None else data[i:i + block]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = c.compress(raw) + c.flush()
            bsize = 18 + len(body) + 8
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize - 1) + body + struct.pack("<II", zlib.crc32(raw), len(raw)))


def test_inflate_threads_give_the_same_records(tmp_path, monkeypatch):
    """Compressed input inflated ahead of the parser (fastq.c: several workers over the blocks of a BGZF file, one gzread thread over any
    other gzip stream, multi-member ones included) against the plain file and against the reader with the threads off: same chunks."""
    rng = random.Random(11)
    a, b = make_text(rng, 6000), make_text(rng, 6000, multiline=True)
    plain1, plain2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq")
    open(plain1, "w").write(a); open(plain2, "w").write(b)
    gz1, gz2 = str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq.gz")
    gzip.open(gz1, "wt").write(a)
    with open(gz2, "wb") as f:   # three members
        k = len(b) // 3
        for part in (b[:k], b[k:2 * k], b[2 * k:]):
            f.write(gzip.compress(part.encode()))
    bg1, bg2 = str(tmp_path / "a.bgz.fq.gz"), str(tmp_path / "b.bgz.fq.gz")
    write_bgzf(bg1, a.encode()); write_bgzf(bg2, b.encode(), block=65000)
    assert gzip.open(bg1, "rt").read() == a          # (zlib itself reads what write_bgzf wrote)
    for chunk in (5000, 200000):
        want = read_all(plain1, plain2, chunk)
        for p1, p2 in ((gz1, gz2), (bg1, bg2), (bg1, plain2)):
            for threads in ("3", "1", "0"):
                monkeypatch.setenv("BSX_INFLATE_THREADS", threads)
                assert read_all(p1, p2, chunk) == want, (p1, p2, threads)
                assert read_all_threaded(p1, p2, chunk) == [[r + (i,) for i, r in enumerate(ch)] for ch in want]
    # a truncated BGZF file ends the input at the damage instead of hanging
    monkeypatch.setenv("BSX_INFLATE_THREADS", "3")
    cut = str(tmp_path / "cut.fq.gz")
    open(cut, "wb").write(open(bg1, "rb").read()[:30000])
    got = read_all(cut, None, 10 ** 7)
    assert 0 < sum(len(c) for c in got) < 6000


def test_skip_chunk_counts_what_the_reader_reads(tmp_path):
    """What a rank does with the chunks of the other ranks over compressed input: bsx_fq_skip_chunk walks a chunk without building records and
    leaves the stream where the next chunk starts -- alternating skipped and parsed chunks gives every other chunk of the plain reader."""
    L = B.lib()
    L.bsx_hook_fq_skip_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bsx_hook_fq_open.restype = C.c_void_p
    L.bsx_hook_fq_open.argtypes = [C.c_char_p]
    L.bsx_hook_fq_close.argtypes = [C.c_void_p]
    L.bsx_hook_fq_chunk.restype = C.POINTER(B.Read)
    L.bsx_hook_fq_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    rng = random.Random(12)
    a, b = make_text(rng, 4000, multiline=True), make_text(rng, 4000, crlf=True)
    p1, p2 = str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq.gz")
    write_bgzf(p1, a.encode()); gzip.open(p2, "wt", newline="").write(b)
    for chunk in (3000, 40000):
        want = read_all(p1, p2, chunk)
        for mine in (0, 1):
            f1, f2 = L.bsx_hook_fq_open(p1.encode()), L.bsx_hook_fq_open(p2.encode())
            k = 0
            while True:
                if k % 2 != mine:
                    n = L.bsx_hook_fq_skip_chunk(f1, f2, chunk)
                    if n == 0:
                        break
                    assert n == len(want[k]), (chunk, k)
                else:
                    n = C.c_int()
                    r = L.bsx_hook_fq_chunk(f1, f2, chunk, 0, C.byref(n))
                    if not r or n.value == 0:
                        break
                    assert [r[i].name.decode() for i in range(n.value)] == [x[0] for x in want[k]]
                    L.bsx_sim_free_reads(r, n.value)
                k += 1
            assert k == len(want)
            L.bsx_hook_fq_close(f1); L.bsx_hook_fq_close(f2)


def _read_names(path):
    L = B.lib()
    L.bsx_hook_fq_open.restype = C.c_void_p
    L.bsx_hook_fq_open.argtypes = [C.c_char_p]
    L.bsx_hook_fq_close.argtypes = [C.c_void_p]
    L.bsx_hook_fq_error.argtypes = [C.c_void_p]
    L.bsx_hook_fq_chunk.restype = C.POINTER(B.Read)
    L.bsx_hook_fq_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    f = L.bsx_hook_fq_open(path.encode())
    assert f
    names = []
    while True:
        n = C.c_int()
        r = L.bsx_hook_fq_chunk(f, None, 10 ** 6, 0, C.byref(n))
        if not r or n.value == 0:
            break
        names += [r[i].name.decode() for i in range(n.value)]
        L.bsx_sim_free_reads(r, n.value)
    err = L.bsx_hook_fq_error(f)
    L.bsx_hook_fq_close(f)
    return names, err


def test_pipes_are_opened_once(tmp_path, monkeypatch):
    """`align ref <(zcat r1.gz) <(zcat r2.gz)`: a FIFO gives every byte once.  The reader must not sniff it with one handle and read it with
    another (round 4: 921 of 1000 reads of a piped plain file, none of a piped gzip file, no error).  Plain and gzipped text through a named
    FIFO and through /dev/fd/N, with and without the inflate threads."""
    import os
    import threading
    rng = random.Random(21)
    text = make_text(rng, 1000)
    open(str(tmp_path / "plain.fq"), "w").write(text)
    want, err0 = _read_names(str(tmp_path / "plain.fq"))   # (names as the reader trims them, bwa.c:58-63)
    assert len(want) == 1000 == len(kseq_records(text)) and err0 == 0
    payloads = {"plain": text.encode(), "gzip": gzip.compress(text.encode())}
    bg = str(tmp_path / "x.bgz")
    write_bgzf(bg, text.encode())
    payloads["bgzf"] = open(bg, "rb").read()
    for threads in ("3", "0"):
        monkeypatch.setenv("BSX_INFLATE_THREADS", threads)
        for kind, data in payloads.items():
            fifo = str(tmp_path / ("fifo_%s_%s" % (kind, threads)))
            os.mkfifo(fifo)

            def feed(path=fifo, data=data):
                with open(path, "wb") as w:
                    w.write(data)
            t = threading.Thread(target=feed)
            t.start()
            names, err = _read_names(fifo)
            t.join()
            assert names == want and err == 0, (kind, threads, len(names))
            # an inherited descriptor of a pipe, named the way a shell's process substitution names it
            rd, wr = os.pipe()
            t = threading.Thread(target=lambda wr=wr, data=data: (os.write(wr, data[:1]), os.write(wr, data[1:]), os.close(wr)))
            t.start()
            names, err = _read_names("/dev/fd/%d" % rd)
            t.join()
            os.close(rd)
            assert names == want and err == 0, (kind, threads, "fd")


def test_damaged_compressed_input_ends_the_stream_and_is_reported(tmp_path, monkeypatch):
    """One flipped byte in the middle of a BGZF file (round 4: the block was skipped and reading went on behind it -- with paired files every
    later R1 met the wrong R2 -- and the exit code stayed 0): the stream ends AT the damage, the reader says so (bsx_fq_error), and a block
    whose CRC32 or ISIZE disagree with what it inflates to counts as damaged.  Truncated files likewise, for every way of reading."""
    rng = random.Random(22)
    text = make_text(rng, 4000)
    open(str(tmp_path / "plain.fq"), "w").write(text)
    want, _ = _read_names(str(tmp_path / "plain.fq"))
    assert len(want) == 4000
    bg = str(tmp_path / "a.fq.gz")
    write_bgzf(bg, text.encode(), block=20000)
    raw = bytearray(open(bg, "rb").read())
    # where the blocks start
    offs, o = [], 0
    while o < len(raw):
        offs.append(o)
        o += (raw[o + 16] | raw[o + 17] << 8) + 1
    assert len(offs) > 6
    k = len(offs) // 2
    monkeypatch.setenv("BSX_INFLATE_THREADS", "3")
    # (a) a byte of the deflate stream, (b) the CRC32, (c) the ISIZE of block k
    for what, at in (("stream", offs[k] + 18 + 40), ("crc", offs[k + 1] - 8), ("isize", offs[k + 1] - 4)):
        bad = bytearray(raw)
        bad[at] ^= 0x5a
        p = str(tmp_path / ("bad_%s.fq.gz" % what))
        open(p, "wb").write(bytes(bad))
        names, err = _read_names(p)
        assert err != 0, what
        assert 0 < len(names) < len(want) and names == want[:len(names)], (what, len(names))   # a prefix: nothing from behind the damage
        # the text of the blocks before k holds at least len(names) - 1 whole records
        assert len(names) <= sum(1 for _ in kseq_records(text[:20000 * k + 20000]))
    names, err = _read_names(bg)
    assert names == want and err == 0
    # truncation: BGZF workers, the gzread thread (plain gzip), and no threads at all
    gz = str(tmp_path / "b.fq.gz")
    open(gz, "wb").write(gzip.compress(text.encode()))
    for src, threads in ((bg, "3"), (gz, "3"), (gz, "0"), (bg, "0")):
        monkeypatch.setenv("BSX_INFLATE_THREADS", threads)
        cut = str(tmp_path / "cut.fq.gz")
        data = open(src, "rb").read()
        open(cut, "wb").write(data[:len(data) * 2 // 3])
        names, err = _read_names(cut)
        assert err != 0 and 0 < len(names) < len(want) and names == want[:len(names)], (src, threads, err, len(names))


def test_command_line_fails_on_damaged_input(tmp_path):
    """the CLI (cli.c, here over the CPU restatement of the kernels: oracle_align) turns the reader's sticky error into exit code 1"""
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import simdata
    from biscuit_amd.api import Index
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cpu = os.path.join(root, "oracle", "oracle_align")
    d = str(tmp_path)
    contigs = simdata.make_genome(60000, seed=5, n_contigs=1)
    simdata.write_genome(d + "/g.fa", contigs)
    Index.build(d + "/g.fa", d + "/g").close()
    simdata.write_fastq(d + "/ok.fq", simdata.make_single(contigs, 400, 100, 3))
    text = open(d + "/ok.fq").read()
    write_bgzf(d + "/ok.fq.gz", text.encode(), block=8000)
    raw = bytearray(open(d + "/ok.fq.gz", "rb").read())
    first = (raw[16] | raw[17] << 8) + 1
    raw[first + 18 + 30] ^= 0x33   # inside the second block
    open(d + "/bad.fq.gz", "wb").write(bytes(raw))
    ok = subprocess.run([cpu, "-@", "2", "g", "ok.fq.gz"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    bad = subprocess.run([cpu, "-@", "2", "g", "bad.fq.gz"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert ok.returncode == 0 and ok.stdout.count(b"\n") > 400
    assert bad.returncode == 1 and b"damaged" in bad.stderr, (bad.returncode, bad.stderr[-600:])
    n_ok = sum(1 for l in ok.stdout.split(b"\n") if l and not l.startswith(b"@"))
    n_bad = sum(1 for l in bad.stdout.split(b"\n") if l and not l.startswith(b"@"))
    assert 0 < n_bad < n_ok
    # and the everyday invocation with a process substitution gives the whole SAM
    ps = subprocess.run(["bash", "-c", "%s -@ 2 g <(zcat ok.fq.gz)" % cpu], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert ps.returncode == 0
    strip = lambda b: [l for l in b.split(b"\n") if not l.startswith(b"@PG")]
    assert strip(ps.stdout) == strip(ok.stdout)
