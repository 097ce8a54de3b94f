"""Multi-GPU `biscuit align`: one process per GPU, chunks of the input are the shard unit.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m biscuit_amd.multi_gpu [--out FILE [--via-rank0]] [--shard chunks|pairs] -- [biscuit align options] <index base> <in1.fq> [in2.fq]

Every rank cuts the input into the reference's chunks (10 Mbp x -@, align.c:576) and aligns chunks
r, r+N, ... on its own GPU against its own HBM-resident copy of the index.  The chunk rule is cumulative, so over plain files a
light scan (no records built: csrc/host/fastq.c) finds the chunk boundaries and each rank seeks to and parses only its own chunks;
over compressed or piped input every rank parses everything and drops the chunks of the others (csrc/host/cli.c).  A chunk is the only unit whose reads depend on
each other (per-chunk insert-size statistics), so the SAM equals the single-GPU / CPU output for the same
-@.  The only communication is the streaming gather of the per-chunk SAM text to rank 0
(biscuit_amd/gather.py: sizes, then exactly the payload, point to point -> RCCL over xGMI), overlapped
with the alignment of the following chunks; rank 0 writes chunks in input order as they arrive, and no
rank ever holds more than a few chunks of output.  With --out FILE (one node: the ranks see the same file) nothing but the chunks' sizes is
exchanged: every rank writes its own chunks into the file at their offsets (gather.py, the direct form), so rank 0 is not the funnel of
eight ranks' SAM text; --via-rank0 keeps the gather for such a file too.

--shard pairs (for inputs with fewer chunks than GPUs -- BASELINE configs[1] is two chunks at -@ 16): every rank takes
every chunk and aligns its own slice of the chunk's pairs.  The one step of a chunk that looks at all of its pairs is mem_pestat
(bwamem.c:464-467), and what it computes is a function of the histogram of insert sizes: the ranks add their histograms (an all-reduce
of 2 * max_ins + 1 counters per chunk, csrc/host/region.c: bsx_pes_hist_hook) and each gets the statistics of the whole chunk, so the
SAM is again the single-GPU one.  The slices leave in rank order as chunks k * world + rank of the same gather.

`main(argv, entry=..., use_gpu=...)`: `entry` is the C entry point with the signature of bsx_align_main.
The product always runs bsx_align_main (HIP; no CPU path exists in this package); tests inject another
entry point and the gloo backend from outside (tests/multi_entry_cpu.py).
"""
import ctypes as C
import os
import sys
import threading


def main(argv=None, entry=None, use_gpu=True):
    argv = list(sys.argv[1:] if argv is None else argv)
    out_path = None   # SAM goes to stdout unless --out FILE (libraries such as gloo also print to stdout)
    shard = "chunks"
    via_rank0 = False
    if "--" in argv:
        k = argv.index("--")
        head, argv = argv[:k], argv[k + 1:]
        if "--out" in head:
            out_path = head[head.index("--out") + 1]
        via_rank0 = "--via-rank0" in head
        if "--shard" in head:
            shard = head[head.index("--shard") + 1]
            if shard not in ("chunks", "pairs"):
                raise SystemExit("--shard chunks|pairs")
    import numpy as np
    import torch
    import torch.distributed as dist
    from .gather import ChunkGather
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        if use_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    os.environ["BSX_DEVICE"] = str(local_rank)

    from . import _lib as B
    L = B.lib()
    failed = [0]      # an output or hook failure on this rank: the rounds go on (the other ranks are inside collectives), the exit status says so
    if entry is None:
        entry = L.bsx_align_main
    C.c_int.in_dll(L, "bsx_shard_rank").value = rank
    C.c_int.in_dll(L, "bsx_shard_world").value = world
    C.c_int.in_dll(L, "bsx_shard_mode").value = 1 if (shard == "pairs" and world > 1) else 0
    pes_hook = None
    if shard == "pairs" and world > 1:
        # the insert-size histograms of a chunk's slices, added over the ranks: 80 KB per chunk, from the aligner's own thread while the
        # gather's collectives run on another -- so on a communicator of its own, and on the host (gloo) in GPU runs too
        pes_group = dist.new_group(backend="gloo")
        PES = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int64), C.c_int)

        def pes_sum(ud, hist, nb):
            try:
                a = np.ctypeslib.as_array(hist, shape=(nb,))
                t = torch.from_numpy(a)       # shares the C buffer: the sum lands in place
                dist.all_reduce(t, group=pes_group)
            except Exception as e:
                failed[0] = 1
                sys.stderr.write("[E::multi_gpu] adding the insert-size histograms failed: %r\n" % (e,))
        pes_hook = PES(pes_sum)
        C.c_void_p.in_dll(L, "bsx_pes_hist_hook").value = C.cast(pes_hook, C.c_void_p).value

    # every rank writing its own chunks needs one node (one file system) and a seekable regular file: anything else goes through rank 0
    from .gather import direct_output_ok
    direct = out_path if (out_path and world > 1 and not via_rank0 and direct_output_ok(out_path, world)) else None
    if out_path and world > 1 and not via_rank0 and not direct and rank == 0:
        sys.stderr.write("[M::multi_gpu] %s: not a regular file on a single node -- the records go through rank 0\n" % out_path)
    out = None
    if rank == 0 and not direct:
        out = open(out_path, "wb") if out_path else sys.stdout.buffer
    written = [0]

    def sink(idx, buf):
        if failed[0]:
            return
        try:
            out.write(buf)
            written[0] += len(buf)
        except Exception as e:
            failed[0] = 1
            sys.stderr.write("[E::multi_gpu] writing the SAM failed: %r\n" % (e,))

    dev = torch.device("cuda", local_rank) if (use_gpu and world > 1) else torch.device("cpu")
    G = ChunkGather(rank, world, dev, sink, direct_path=direct)
    HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t)

    def emit(ud, idx, text, n):
        # called on the aligner's writer thread, chunk by chunk in this rank's order; idx -1 = the header (rank 0 only)
        try:
            data = np.ctypeslib.as_array(C.cast(text, C.POINTER(C.c_uint8)), shape=(n,)).copy() if n else np.zeros(0, dtype=np.uint8)
            if idx < 0:
                if direct:
                    G.header = data.tobytes()   # (rank 0 only; before its first chunk: the gather's thread writes it at offset 0)
                else:
                    out.write(data)
                return
        except Exception as e:     # (an exception must not leave a ctypes callback: it would only be printed)
            failed[0] = 1
            sys.stderr.write("[E::multi_gpu] taking a chunk's records failed: %r\n" % (e,))
            data = np.zeros(0, dtype=np.uint8)
            if idx < 0:
                return
        G.submit(idx, data)   # blocks when a few chunks are waiting for their round: back-pressure on the aligner

    hook = HOOK(emit)
    C.c_void_p.in_dll(L, "bsx_emit_hook").value = C.cast(hook, C.c_void_p).value
    args = [b"biscuit_align"] + [a.encode() for a in argv]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    rc_box = [1]

    def work():
        try:
            rc_box[0] = entry(len(args), arr)   # ctypes releases the GIL for the duration of the call
        finally:
            G.close()

    th = threading.Thread(target=work)
    th.start()
    G.run()
    th.join()
    rc = rc_box[0] or failed[0] or (1 if G.failed else 0)
    if world > 1:
        t = torch.tensor([rc], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rc = int(t.item())
    if rank == 0 and out is not None:
        out.flush()
        if out_path:
            out.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
