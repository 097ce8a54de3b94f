"""Streaming gather of per-chunk alignment records to rank 0 ("replicas + gather").

Chunks are dealt round-robin: chunk k is aligned by rank k % world.  The gather runs in rounds; round r
carries chunks r*world .. r*world+world-1, one per rank at most:
    1. every rank reports (has_chunk, n_bytes) -> all ranks (a 2-int all_gather: everybody must see the end),
    2. every rank but 0 that has a chunk sends exactly n_bytes to rank 0 (send/recv: RCCL point-to-point over
       xGMI on GPUs, gloo in the CPU tests); nothing is padded and no other rank receives anything.  Rank 0 posts the
       receives of a round together, each into its own staging buffer: xGMI is point to point, every sender has its own
       link into rank 0, so the round's transfers run side by side instead of one after the other,
    3. rank 0 hands the round's chunks to `sink` in input order and drops them; the device-to-host copies of the later
       senders' chunks (a side stream, pinned buffers) overlap the sink of the earlier ones.
With `direct_path` (the launcher's --out FILE; one node, one file system) no payload moves at all: step 1 carries the sizes to everybody,
every rank derives the offset of its own chunk in the output (header + the chunks before it) and writes it there itself (os.pwrite),
so rank 0 is not the funnel of eight ranks' SAM text.
It ends with the first round in which some rank has no chunk (round-robin dealing: no later chunk exists).
Producers (the aligner's writer thread) hand chunks in through `submit`, which blocks once `max_pending`
chunks wait: host memory of a rank is bounded by a few chunks whatever the input size, rank 0 writes as it
goes, and the transfer of round r overlaps the alignment of the next chunks.
"""
import queue

import torch
import torch.distributed as dist

_EOF = object()


class ChunkGather:
    def __init__(self, rank, world, device, sink, max_pending=3, direct_path=None):
        """device: torch.device the collectives run on (cuda:N under RCCL, cpu under gloo);
        sink(chunk_index, buffer) is called on rank 0 only, in increasing chunk order."""
        self.rank, self.world, self.device, self.sink = rank, world, device, sink
        self.q = queue.Queue(maxsize=max_pending)
        self.bytes_moved = 0          # payload bytes received by rank 0 from other ranks
        self.rounds = 0
        self._stash = {}
        self._pin = None
        self._dev_buf = None
        self._src_buf = {}            # rank 0: per-sender staging (device tensor, pinned host tensor)
        import os
        self._sequential = bool(os.environ.get("BSX_GATHER_SEQUENTIAL"))   # one receive at a time (the form of rounds 1-2)
        self._copy_stream = None
        self.direct_path = direct_path if world > 1 else None   # every rank writes its own chunks into this file at their offsets
        self.header = b""             # rank 0, direct form: the SAM header (set by the producer before its first chunk)
        self.bytes_written = 0        # direct form: bytes this rank wrote itself
        self._dead = False            # set when the gather has ended: late producers (another rank failed) are not blocked
        self.failed = False           # direct form: this rank could not open / write the file; it still takes part in every round

    # ---- producer side -------------------------------------------------------------------------------------------
    def submit(self, chunk_index, data):
        """data: writable 1-d numpy uint8 array (kept by reference until its round is done)"""
        while not self._dead:
            try:
                self.q.put((int(chunk_index), data), timeout=0.2)
                return
            except queue.Full:
                pass

    def close(self):
        """no more chunks from this rank"""
        while not self._dead:
            try:
                self.q.put(_EOF, timeout=0.2)
                return
            except queue.Full:
                pass

    # ---- consumer side (one thread per rank, the one that owns the process group) ------------------------------------
    def _next_own(self, want):
        """this rank's chunk `want`, or None when the producer is done"""
        if want in self._stash:
            return self._stash.pop(want)
        if _EOF in self._stash:
            return None
        while True:
            item = self.q.get()
            if item is _EOF:
                self._stash[_EOF] = True   # stay at end of stream for later rounds
                return None
            k, data = item
            if k == want:
                return data
            self._stash[k] = data     # (producers deliver in order; kept for safety)

    def _staging(self, n):
        """a device (or host, under gloo) byte tensor of at least n bytes, reused from round to round"""
        if self._dev_buf is None or self._dev_buf.numel() < n:
            cap = max(n + (n >> 2), 1 << 20)
            self._dev_buf = torch.empty(cap, dtype=torch.uint8, device=self.device)
            if self.device.type == "cuda":
                self._pin = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        return self._dev_buf[:n]

    def _staging_src(self, src, n):
        """rank 0: sender `src`'s own staging pair, so that the receives of a round can be in flight together"""
        cur = self._src_buf.get(src)
        if cur is None or cur[0].numel() < n:
            cap = max(n + (n >> 2), 1 << 20)
            dev = torch.empty(cap, dtype=torch.uint8, device=self.device)
            pin = torch.empty(cap, dtype=torch.uint8, pin_memory=True) if self.device.type == "cuda" else None
            cur = self._src_buf[src] = (dev, pin)
        return cur[0][:n], (cur[1][:n] if cur[1] is not None else None)

    def warm(self, n_bytes=1 << 20):
        """one empty round before the data: the metadata all_gather and a send from every rank to rank 0, so that the
        point-to-point connections (RCCL sets a pair's channel up on its first transfer) and the staging buffers exist
        before the first chunk is on its way.  Collective: every rank calls it, from the thread that will call run()."""
        if self.world == 1:
            return
        meta = torch.zeros(2, dtype=torch.int64, device=self.device)
        dist.all_gather([torch.zeros(2, dtype=torch.int64, device=self.device) for _ in range(self.world)], meta)
        if self.rank == 0:
            for src in range(1, self.world):   # every sender's own staging pair at its final size, touched once
                dev, pin = self._staging_src(src, n_bytes)
                dist.recv(dev, src=src)
                if pin is not None:
                    pin.copy_(dev)
        else:
            buf = self._staging(n_bytes)
            buf.zero_()
            dist.send(buf, dst=0)
        # Does the backend post the receives of a round together?  Found out here, on 16 bytes per sender, where a backend that fails
        # half-way is attributable to the probe; run() then either groups its receives or takes them one by one, and treats a failure
        # of a grouped round as fatal instead of re-issuing receives that may already be matched.
        if self.rank == 0:
            if not self._sequential:
                ops = [dist.P2POp(dist.irecv, self._staging_src(src, n_bytes)[0][:16], src) for src in range(1, self.world)]
                try:
                    for q in dist.batch_isend_irecv(ops):
                        q.wait()
                except Exception as e:
                    import sys
                    sys.stderr.write("[W::gather] no grouped point-to-point (%r): receives one after the other\n" % (e,))
                    self._sequential = True
                    for src in range(1, self.world):
                        dist.recv(self._staging_src(src, n_bytes)[0][:16], src=src)
            else:
                for src in range(1, self.world):
                    dist.recv(self._staging_src(src, n_bytes)[0][:16], src=src)
        else:
            dist.send(self._staging(n_bytes)[:16], dst=0)

    def run(self):
        """gather until the input ends; returns the number of chunks seen (all ranks return the same)"""
        r = 0
        n_chunks = 0
        if self.direct_path is not None:
            return self._run_direct()
        while True:
            mine = self._next_own(r * self.world + self.rank)
            n = len(mine) if mine is not None else 0
            if self.world == 1:
                if mine is None:
                    break
                self.sink(r, mine)
                n_chunks += 1
                r += 1
                continue
            meta = torch.tensor([1 if mine is not None else 0, n], dtype=torch.int64, device=self.device)
            metas = [torch.zeros(2, dtype=torch.int64, device=self.device) for _ in range(self.world)]
            dist.all_gather(metas, meta)
            metas = [(int(m[0]), int(m[1])) for m in torch.stack(metas).cpu()]
            if self.rank == 0:
                # all receives of the round at once, each into the sender's own buffer
                bufs, ops = {}, []
                for src in range(1, self.world):
                    has, nb = metas[src]
                    if has and nb:
                        bufs[src] = self._staging_src(src, nb)
                        ops.append(dist.P2POp(dist.irecv, bufs[src][0], src))
                reqs = []
                if ops and not self._sequential:
                    try:
                        reqs = dist.batch_isend_irecv(ops)
                    except Exception:   # (warm() probed this form: a failure here may have posted part of the group, so nothing is retried)
                        self._dead = True
                        raise
                if ops and self._sequential:
                    for src in sorted(bufs):
                        dist.recv(bufs[src][0], src=src)
                if metas[0][0]:
                    self.sink(r * self.world, mine)   # this rank's own chunk while the others arrive
                    n_chunks += 1
                for q in reqs:
                    q.wait()
                done = {}
                if self.device.type == "cuda" and bufs:
                    # device -> pinned host on a side stream, sender by sender; the sink of one chunk runs while the next one is copied
                    # (a pageable copy of a chunk's records -- half a GB -- takes longer than the chunk took to align on eight GPUs)
                    if self._copy_stream is None:
                        self._copy_stream = torch.cuda.Stream(device=self.device)
                    self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self._copy_stream):
                        for src in sorted(bufs):
                            dev, pin = bufs[src]
                            pin.copy_(dev, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(self._copy_stream)
                            done[src] = ev
                for src in range(1, self.world):
                    has, nb = metas[src]
                    if not has:
                        continue
                    if nb:
                        dev, pin = bufs[src]
                        if pin is not None:
                            done[src].synchronize()
                            host = pin
                        else:
                            host = dev
                        self.sink(r * self.world + src, memoryview(host.numpy()))
                    else:
                        self.sink(r * self.world + src, memoryview(b""))
                    self.bytes_moved += nb
                    n_chunks += 1
            else:
                n_chunks += sum(1 for has, _ in metas if has)
                if mine is not None and n:
                    buf = self._staging(n)
                    src_t = torch.from_numpy(mine)
                    if self.device.type == "cuda":
                        self._pin[:n].copy_(src_t)
                        buf.copy_(self._pin[:n], non_blocking=True)
                    else:
                        buf.copy_(src_t)
                    dist.send(buf, dst=0)
            self.rounds += 1
            r += 1
            if not all(has for has, _ in metas):
                break
        self._dead = True
        return n_chunks

    def _run_direct(self):
        """the rounds of run() with the sizes alone: every rank writes its own chunk at header + (bytes of all chunks before it).
        A rank that cannot open or write the file (ENOSPC, a target that is not seekable ...) says so once, sets `failed` and keeps taking
        part in the size exchanges: the other ranks are inside the same collectives, and the launcher's MAX-reduced exit status reports it."""
        import os
        import sys
        r = 0
        n_chunks = 0
        base = 0
        fd = -1

        def fail(what, e):
            if not self.failed:
                sys.stderr.write("[E::gather] rank %d: %s %s failed: %r\n" % (self.rank, what, self.direct_path, e))
            self.failed = True
        try:
            while True:
                mine = self._next_own(r * self.world + self.rank)
                n = len(mine) if mine is not None else 0
                hdr = 0
                if r == 0 and self.rank == 0:   # the file exists, with its header, before anybody learns the header's length
                    hdr = len(self.header)
                    try:
                        fd = os.open(self.direct_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                        if hdr:
                            os.pwrite(fd, bytes(self.header), 0)
                    except OSError as e:
                        fail("opening", e)
                meta = torch.tensor([1 if mine is not None else 0, n, hdr], dtype=torch.int64, device=self.device)
                metas = [torch.zeros(3, dtype=torch.int64, device=self.device) for _ in range(self.world)]
                dist.all_gather(metas, meta)
                metas = [(int(m[0]), int(m[1]), int(m[2])) for m in torch.stack(metas).cpu()]
                if r == 0:
                    base = metas[0][2]
                    if self.rank != 0:
                        try:
                            fd = os.open(self.direct_path, os.O_WRONLY)
                        except OSError as e:
                            fail("opening", e)
                off = base + sum(nb for has, nb, _ in metas[:self.rank] if has)
                if mine is not None and n and not self.failed:
                    view = memoryview(mine)
                    done = 0
                    try:
                        while done < n:
                            done += os.pwrite(fd, view[done:], off + done)
                        self.bytes_written += n
                    except OSError as e:
                        fail("writing", e)
                base += sum(nb for has, nb, _ in metas if has)
                n_chunks += sum(1 for has, _, _ in metas if has)
                self.rounds += 1
                r += 1
                if not all(has for has, _, _ in metas):
                    break
        finally:
            self._dead = True
            if fd >= 0:
                try:
                    os.close(fd)
                except OSError as e:
                    fail("closing", e)
        return n_chunks


def direct_output_ok(path, world, env=None):
    """May every rank write its own chunks into `path` (ChunkGather(direct_path=...))?  Only when all ranks are on one node -- they must see
    the same file -- and the target is (or will be created as) a regular file: pwrite() at offsets needs a seekable file, which /dev/stdout,
    a FIFO or a character device is not.  The answer is the same on every rank of a one-node job."""
    import os
    import stat
    env = os.environ if env is None else env
    try:
        if int(env.get("LOCAL_WORLD_SIZE", "0")) != int(world):
            return False
    except ValueError:
        return False
    try:
        st = os.stat(path)
    except FileNotFoundError:
        return os.path.isdir(os.path.dirname(os.path.abspath(path)))
    except OSError:
        return False
    return stat.S_ISREG(st.st_mode)
