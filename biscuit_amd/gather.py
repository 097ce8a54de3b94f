"""Streaming gather of per-chunk alignment records to rank 0 ("replicas + gather").

Chunks are dealt round-robin: chunk k is aligned by rank k % world.  The gather runs in rounds; round r
carries chunks r*world .. r*world+world-1, one per rank at most:
    1. every rank reports (has_chunk, n_bytes) -> all ranks (a 2-int all_gather: everybody must see the end),
    2. every rank but 0 that has a chunk sends exactly n_bytes to rank 0 (send/recv: RCCL point-to-point over
       xGMI on GPUs, gloo in the CPU tests); nothing is padded and no other rank receives anything,
    3. rank 0 hands the round's chunks to `sink` in input order and drops them.
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
    def __init__(self, rank, world, device, sink, max_pending=3):
        """device: torch.device the collectives run on (cuda:N under RCCL, cpu under gloo);
        sink(chunk_index, buffer) is called on rank 0 only, in increasing chunk order."""
        self.rank, self.world, self.device, self.sink = rank, world, device, sink
        self.q = queue.Queue(maxsize=max_pending)
        self.bytes_moved = 0          # payload bytes received by rank 0 from other ranks
        self.rounds = 0
        self._stash = {}
        self._pin = None
        self._dev_buf = None
        self._dead = False            # set when the gather has ended: late producers (another rank failed) are not blocked

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

    def warm(self, n_bytes=1 << 20):
        """one empty round before the data: the metadata all_gather and a send from every rank to rank 0, so that the
        point-to-point connections (RCCL sets a pair's channel up on its first transfer) and the staging buffers exist
        before the first chunk is on its way.  Collective: every rank calls it, from the thread that will call run()."""
        if self.world == 1:
            return
        meta = torch.zeros(2, dtype=torch.int64, device=self.device)
        dist.all_gather([torch.zeros(2, dtype=torch.int64, device=self.device) for _ in range(self.world)], meta)
        buf = self._staging(n_bytes)
        if self.rank == 0:
            for src in range(1, self.world):
                dist.recv(buf, src=src)
            if self.device.type == "cuda":
                self._pin[:n_bytes].copy_(buf)
        else:
            buf.zero_()
            dist.send(buf, dst=0)

    def run(self):
        """gather until the input ends; returns the number of chunks seen (all ranks return the same)"""
        r = 0
        n_chunks = 0
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
                for src in range(self.world):
                    has, nb = metas[src]
                    if not has:
                        continue
                    if src == 0:
                        self.sink(r * self.world, mine)
                    else:
                        buf = self._staging(nb)
                        if nb:
                            dist.recv(buf, src=src)
                        if self.device.type == "cuda":   # through the pinned staging buffer: a pageable copy of a chunk's records
                            host = self._pin[:nb]        # (half a GB) takes longer than the chunk took to align on eight GPUs
                            host.copy_(buf)
                        else:
                            host = buf
                        self.sink(r * self.world + src, memoryview(host.numpy()))
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
