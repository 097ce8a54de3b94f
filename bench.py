#!/usr/bin/env python3
"""bench.py -- paired-end reads aligned per second through the MI355X `biscuit align` hot path.

A step = one chunk (10 Mbp x threads of 2x150 bp reads, the reference's chunk size, align.c:576)
pushed through bsx_process_seqs (== mem_process_seqs): all five HIP kernels + the host stages
between them, ending in SAM text.  The index is resident in HBM before timing starts; reads are in
host memory as bseq1_t records, exactly what mem_process_seqs receives.
hg38 is not available offline: the genome is a seeded synthetic one (size --genome-mbp), and the
JSON line says so.  N>1: one process per GPU (torchrun), chunks are the shard unit (every chunk is
independent in the reference, so the output equals the single-GPU run); NCCL(=RCCL) carries only the
barrier and the final reduction of counts/timings.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("BSX_BENCH_GENOME_MBP", "3100")))
    ap.add_argument("--threads", type=int, default=16, help="-@ of the run: fixes the chunk size (10 Mbp x threads), like the reference")
    ap.add_argument("--host-threads", type=int, default=0, help="worker threads for the host stages; 0 = cores / ranks, capped at 64")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample-pairs", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one chunk at a time through bsx_process_seqs (no overlap of consecutive chunks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from biscuit_amd import _lib as B
    from biscuit_amd.api import Index, Device, default_opt
    L = B.lib()   # raises if libbiscuit_amd.so has not been built

    ncores = effective_cores()
    threads = max(1, args.threads)
    # the host pool is slightly oversubscribed: its workers block on the device batches and on each other (measured 16 -> 24
    # threads on 16 cores: +2 %)
    host_threads = args.host_threads if args.host_threads > 0 else max(1, min(96, (ncores // max(1, world)) * 3 // 2))
    os.environ["BSX_HOST_THREADS"] = str(host_threads)
    n_bases = int(args.genome_mbp * 1e6)

    def barrier():
        if dist is not None:
            dist.barrier()

    # genome + both FM indices: generated (seeded) and indexed on this rank's own GPU (csrc/hip/k_index.hip; an hg38-sized
    # genome takes ~5 s to generate and ~13 s to index), resident in HBM from then on.  Rank 0 of a single-GPU run also
    # takes the file-format arrays back to the host: the CPU baseline needs them.
    t0 = time.time()
    idx = Index.synthetic(n_bases, seed=2024, n_contigs=24 if n_bases >= 1_000_000_000 else 8)
    dev = Device(local_rank)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    dev.build_index(idx, fill_host=want_cpu)
    t_build = time.time() - t0
    barrier()

    opt = default_opt()
    opt.n_threads = threads
    # defaults: -b 0 (non-directional search: 4 strand searches per pair)
    opt.flag |= 0x10 | 0x2            # MEM_F_NO_MULTI (align.c:335) | MEM_F_PE
    pairs_per_step = (opt.chunk_size * threads) // (2 * args.read_len)   # the reference's chunk: 10 Mbp x threads
    n_reads = pairs_per_step * 2

    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_sam_bytes.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_sam_bytes.restype = C.c_int64

    def gen(seed, n_pairs):
        p = C.c_void_p()
        B.check(L.bsx_sim_pairs(idx.h, n_pairs, args.read_len, seed, 200, 500, 0.005, 0.0, C.byref(p)), "sim_pairs")
        return p

    C.c_int.in_dll(L, "bsx_verbose").value = 1   # silence per-chunk messages inside the timed region

    # a bounded set of distinct chunks, reused round-robin (a chunk is only pushed again long after it has completed and its
    # SAM text has been dropped): keeps the host memory of a rank at a few GB whatever --steps is
    class Ring(list):
        def __getitem__(self, i):
            return list.__getitem__(self, i % len(self))
    chunks = Ring(gen(1000 * (rank + 1) + s, pairs_per_step) for s in range(min(args.warmup + args.steps, 8)))
    # chunks go through the two-deep pipeline of include/bsx.h (front half of chunk k+1 on the device while the host
    # finishes chunk k); --no-pipeline runs them one at a time through bsx_process_seqs instead
    L.bsx_stream_open.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.bsx_stream_push.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    L.bsx_stream_flush.argtypes = [C.c_void_p]
    L.bsx_stream_close.argtypes = [C.c_void_p]
    L.bsx_stream_close.restype = None
    stream = C.c_void_p()
    depth = 1
    if not args.no_pipeline:
        B.check(L.bsx_stream_open(dev.h, C.byref(opt), idx.h, None, C.byref(stream)), "stream_open")
        L.bsx_stream_depth.argtypes = [C.c_void_p]
        depth = L.bsx_stream_depth(stream)
    n_processed = 0
    for s in range(args.warmup):
        if args.no_pipeline:
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n_processed, n_reads, chunks[s], None), "process_seqs(warmup)")
        else:
            B.check(L.bsx_stream_push(stream, n_processed, n_reads, chunks[s]), "stream_push(warmup)")
        n_processed += n_reads
    if not args.no_pipeline:
        B.check(L.bsx_stream_flush(stream), "stream_flush(warmup)")
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]
    for s_ in range(args.warmup):
        L.bsx_sim_reset_reads(chunks[s_], n_reads)   # drop the warm-up chunks' SAM text
    for k in range(8):
        dev.kernel_time(k, reset=True)
    dev.counters(reset=True)
    phase_tot = {}
    barrier()
    torch.cuda.synchronize()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.time()
    sam_bytes_box = [0]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]

    # A completed chunk's SAM text is counted and dropped (13 chunks of SAM would be several GB per rank) by a consumer
    # thread, as the command line's writer thread does (cli.c): the stream runs a chunk's back half on the thread that
    # pushes, so consuming the output there would stall the pipeline.  The consumer is joined inside the timed region.
    import queue
    import threading
    retire_q = queue.Queue()
    retire_s = [0.0]
    retired = set()

    def consumer():
        while True:
            k = retire_q.get()
            if k is None:
                return
            tr = time.time()
            sam_bytes_box[0] += L.bsx_sim_sam_bytes(chunks[k], n_reads)
            L.bsx_sim_reset_reads(chunks[k], n_reads)
            retire_s[0] += time.time() - tr
            retired.add(k)
    consumer_th = threading.Thread(target=consumer)
    consumer_th.start()

    def retire(k):
        retire_q.put(k)

    def account():
        ps = B.PhaseStats()
        L.bsx_last_phase_stats(C.byref(ps))
        for f, _ in B.PhaseStats._fields_:
            phase_tot[f] = phase_tot.get(f, 0) + getattr(ps, f)
        phase_tot["_chunks"] = phase_tot.get("_chunks", 0) + 1

    for s in range(args.warmup, args.warmup + args.steps):
        if args.no_pipeline:
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n_processed, n_reads, chunks[s], None), "process_seqs")
            account()
            retire(s)
        else:
            while s - len(chunks) >= args.warmup and (s - len(chunks)) not in retired:
                time.sleep(0.001)   # the ring slot's previous use must have been consumed (it has, several steps ago)
            B.check(L.bsx_stream_push(stream, n_processed, n_reads, chunks[s]), "stream_push")
            if s - args.warmup >= depth - 1:
                account()       # the push completed the chunk pushed depth-1 pushes ago
                retire(s - (depth - 1))
        n_processed += n_reads
    if not args.no_pipeline:
        B.check(L.bsx_stream_flush(stream), "stream_flush")   # the chunks still in flight complete inside the timed region
        account()               # (the statistics of the last one stand in for the others drained with it)
        for k in range(max(args.warmup, args.warmup + args.steps - (depth - 1)), args.warmup + args.steps):
            retire(k)
    retire_q.put(None)
    consumer_th.join()
    torch.cuda.synchronize()
    barrier()
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    sam_bytes = sam_bytes_box[0]

    tmax, tot_reads = dt, n_reads * args.steps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        c = torch.tensor([tot_reads], dtype=torch.int64, device="cuda")
        dist.all_reduce(c)   # "gather" of per-GPU record counts over RCCL
        tot_reads = int(c.item())

    # rooflines of the two HBM-bound kernels, both random 64-byte gathers over the FM index:
    #   k_seed (K1+K2): 64 B per FM block touched by bwt_extend (two blocks unless k and l share one)
    #   k_occ  (K3):    64 B per LF step of bwt_sa + 8 B per SA sample + 16 B per occurrence (rank in, position out)
    # Durations are HIP-event times on the launch stream over the timed region.  With chunks pipelined, kernels of
    # different chunks share the device, so a launch lasts longer than it would alone; one extra chunk is therefore
    # run unpipelined after the timed region and its event times are reported next to the live ones.
    ctr = dev.counters()
    ktimes = [dev.kernel_time(k) for k in range(8)]
    alone = None
    if not args.no_pipeline:
        for k in range(8):
            dev.kernel_time(k, reset=True)
        extra = gen(777, pairs_per_step)
        B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n_processed, n_reads, extra, None), "process_seqs(standalone)")
        L.bsx_sim_free_reads(extra, n_reads)
        alone = [dev.kernel_time(k) for k in range(8)]

    def roof_of(name, k, alg_bytes, extra):
        ms, launches = ktimes[k]
        if not launches or ms <= 0:
            return None
        ach = alg_bytes / (ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5),
             "traffic": None, "algorithmic_bytes_per_launch": alg_bytes / launches, "avg_launch_ms": round(ms / launches, 3)}
        if alone and alone[k][1]:
            r["avg_launch_ms_standalone"] = round(alone[k][0] / alone[k][1], 3)
            r["achieved_standalone"] = round(alg_bytes / launches / (alone[k][0] / alone[k][1] * 1e-3) / 1e9, 2)
        r.update(extra)
        return r

    # HBM traffic per launch from the committed counter passes (rocprofv3 --pmc cannot run inside this process; the
    # passes profile this same command, one chunk per launch), only quoted when the workload is the profiled one
    traffic = {}
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    if os.path.exists(tpath) and abs(args.genome_mbp - 128) < 1e-9 and args.read_len == 150 and threads == 16:
        with open(tpath) as f:
            tj = json.load(f)
        if tj.get("_reads_per_chunk") == n_reads:   # per launch = per chunk
            for kname in ("k_seed", "k_occ"):
                if tj.get(kname, {}).get("FETCH_SIZE_KiB") is not None and tj[kname].get("WRITE_SIZE_KiB") is not None:
                    traffic[kname] = 1024.0 * (tj[kname]["FETCH_SIZE_KiB"] + tj[kname]["WRITE_SIZE_KiB"])

    roof = roof_of("k_seed (K1+K2 SMEM seeding)", 0, 64.0 * (ctr[0] + ctr[1]),
                   {"fm_block_touches_per_read": (ctr[0] + ctr[1]) / float(n_reads * args.steps)})
    roof_other = roof_of("k_occ (K3 suffix-array lookups of the whole chunk)", 1, 64.0 * ctr[2] + 24.0 * ctr[3],
                         {"lf_steps_per_read": ctr[2] / float(n_reads * args.steps), "sa_lookups_per_read": ctr[3] / float(n_reads * args.steps)})
    for r, kname in ((roof, "k_seed"), (roof_other, "k_occ")):
        if r and kname in traffic:
            r["traffic"] = traffic[kname]
            r["traffic_source"] = "FETCH_SIZE+WRITE_SIZE of profiles/r01_traffic.json: separate rocprofv3 --pmc passes of this command, calibrated on k_occ's known 64-byte gathers"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(L, B, idx, opt, args, ncores)

    if rank == 0:
        names = ["seed", "occ", "extend", "sw", "global", "regions_tier1", "regions_tiers23", "seed_host_path_batches"]
        out = {
            "metric": "paired-end reads aligned/sec", "value": round(tot_reads / tmax, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * tmax / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] shape: 2x%d bp synthetic directional bisulfite pairs vs a SYNTHETIC %.0f Mbp genome with repeat families "
                                   "(hg38 itself is not available offline; SURVEY 8(d) config 2 fallback: an hg38-sized synthetic genome, two FM indices of %.2f G symbols each, "
                                   "built on the GPU at start-up), biscuit align defaults (-b 0)" % (args.read_len, args.genome_mbp, 2 * n_bases / 1e9),
                       "reads_per_step_per_gpu": n_reads, "chunk_threads(-@)": threads, "host_threads_per_gpu": host_threads, "host_cores_usable": ncores, "parallelism": "chunk-sharded x%d" % world, "chunk_pipeline_depth": depth,
                       "index_bytes_in_hbm": int(2 * (n_bases * 2 / 128 * 64 + n_bases * 2 / 4 * 8) + n_bases / 4)},
            "roofline": roof,
            "roofline_second_kernel": roof_other,
            "cpu_baseline": cpu,
            "kernel_ms_per_step": {names[k]: round(ktimes[k][0] / args.steps, 3) for k in range(8)},
            "kernel_ms_per_step_standalone": ({names[k]: round(alone[k][0], 3) for k in range(8)} if alone else None),
            "strand_searches_per_step": phase_tot.get("n_tasks", 0) // max(1, phase_tot.get("_chunks", 1)), "strand_searches_chained_on_host_per_step": phase_tot.get("n_host_tasks", 0) // max(1, phase_tot.get("_chunks", 1)),
            "host_phase_s_per_chunk": {k: round(v / max(1, phase_tot.get("_chunks", 1)), 4) for k, v in phase_tot.items() if k.startswith("t_")},
            "sam_consumer_s_per_step": round(retire_s[0] / args.steps, 4),
            "host_cpu_s_per_step": {"user": round((ru1.ru_utime - ru0.ru_utime) / args.steps, 2), "system": round((ru1.ru_stime - ru0.ru_stime) / args.steps, 2)},
            "sam_bytes_per_read": round(sam_bytes / float(n_reads * args.steps), 1),
            "genome_and_index_build_s": round(t_build, 1), "device": dev.name,
        }
        print(json.dumps(out))
    if not args.no_pipeline:
        L.bsx_stream_close(stream)
    for c in list.__iter__(chunks):
        L.bsx_sim_free_reads(c, n_reads)
    if dist is not None:
        dist.destroy_process_group()


def effective_cores():
    """CPUs this process may actually use: min(online, affinity mask, cgroup quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, int(q / p + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(L, B, idx, opt, args, ncores):
    """The CPU restatement (oracle/: same host pipeline over scalar C kernels, pthreads) on a bounded
    sample of the same workload, all host cores.  kind = "port": the full reference cannot be built
    offline (memchain.c & co. need un-vendored headers), see DESIGN.md."""
    import oracle_lib
    port = oracle_lib.Port(idx, n_threads=ncores)
    be = port.backend()
    o = B.Opt.from_buffer_copy(opt)
    os.environ["BSX_HOST_THREADS"] = str(ncores)
    n_pairs = args.cpu_sample_pairs
    p = C.c_void_p()
    B.check(L.bsx_sim_pairs(idx.h, n_pairs, args.read_len, 999, 200, 500, 0.005, 0.0, C.byref(p)), "sim_pairs")
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    t0 = time.time()
    B.check(L.bsx_process_seqs_backend(C.byref(be), C.byref(o), idx.h, 0, n_pairs * 2, p, None), "cpu baseline")
    dt = time.time() - t0
    L.bsx_sim_free_reads(p, n_pairs * 2)
    return {"value": round(n_pairs * 2 / dt, 1), "unit": "reads/s", "cores": ncores, "kind": "port",
            "sample": "%d pairs of the same workload, one chunk, %.1f s" % (n_pairs, dt)}


if __name__ == "__main__":
    main()
