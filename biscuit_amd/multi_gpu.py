"""Multi-GPU `biscuit align`: one process per GPU, chunks of the input are the shard unit.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m biscuit_amd.multi_gpu [--backend hip|oracle] [--out FILE] -- [biscuit align options] <index base> <in1.fq> [in2.fq]

Every rank streams the same FASTQ(s) and cuts them into the reference's chunks (10 Mbp x -@, align.c:576);
rank r aligns chunks r, r+N, ... on its own GPU against its own HBM-resident copy of the index.  A chunk
is the only unit whose reads depend on each other (per-chunk insert-size statistics), so the SAM equals
the single-GPU / CPU output for the same -@.  The only communication is the gather of the per-chunk SAM
text to rank 0 (sizes, then padded bytes: torch.distributed all_gather -> RCCL over xGMI on GPUs, gloo
for the CPU tests), which writes the chunks back in input order.
"""
import ctypes as C
import os
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    backend = "hip"
    out_path = None   # SAM goes to stdout unless --out FILE (libraries such as gloo also print to stdout)
    if "--" in argv:
        k = argv.index("--")
        head, argv = argv[:k], argv[k + 1:]
        if "--backend" in head:
            backend = head[head.index("--backend") + 1]
        if "--out" in head:
            out_path = head[head.index("--out") + 1]
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = backend == "hip"
    if world > 1:
        if use_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    os.environ["BSX_DEVICE"] = str(local_rank)

    from . import _lib as B
    L = B.lib()
    entry = L.bsx_align_main
    if backend == "oracle":   # tests only: the CPU restatement under oracle/
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        entry = C.CDLL(os.path.join(root, "oracle", "liboracle_port.so")).oracle_align_main
    C.c_int.in_dll(L, "bsx_shard_rank").value = rank
    C.c_int.in_dll(L, "bsx_shard_world").value = world
    chunks = {}
    HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t)

    def emit(ud, idx, text, n):
        chunks[int(idx)] = C.string_at(text, n)

    hook = HOOK(emit)
    C.c_void_p.in_dll(L, "bsx_emit_hook").value = C.cast(hook, C.c_void_p).value
    args = [b"biscuit_align"] + [a.encode() for a in argv]
    arr = (C.c_char_p * (len(args) + 1))(*args, None)
    rc = entry(len(args), arr)

    # gather the per-chunk records on rank 0
    keys = sorted(chunks)
    blob = b"".join(chunks[k] for k in keys)
    meta = [(k, len(chunks[k])) for k in keys]
    if world > 1:
        dev = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
        metas = [None] * world
        dist.all_gather_object(metas, (rc, meta))
        sizes = [sum(n for _, n in m[1]) for m in metas]
        pad = max(sizes + [1])
        mine = torch.zeros(pad, dtype=torch.uint8, device=dev)
        if blob:
            mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
        got = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(got, mine)
        if rank == 0:
            rc = max(m[0] for m in metas)
            parts = {}
            for r in range(world):
                buf = got[r].cpu().numpy().tobytes()
                at = 0
                for k, n in metas[r][1]:
                    parts[k] = buf[at:at + n]
                    at += n
            out = open(out_path, "wb") if out_path else sys.stdout.buffer
            for k in sorted(parts):
                out.write(parts[k])
            out.flush()
        dist.barrier()
        dist.destroy_process_group()
    else:
        out = open(out_path, "wb") if out_path else sys.stdout.buffer
        for k in keys:
            out.write(chunks[k])
        out.flush()
    return rc


if __name__ == "__main__":
    sys.exit(main())
