#!/usr/bin/env python3
"""bench.py -- paired-end reads aligned per second through the MI355X `biscuit align` hot path.

Since round 5 the headline workload is the hg38-LIKE synthetic genome (BASELINE's metric is quoted "vs hg38": the repeat families of a
real genome are where max_occ binds, strand searches have thousands of seeds and a read has dozens of regions); the clean genome of rounds
1-4, 1 kb single-end reads (configs[4]) and the command line end to end are sub-records of the same JSON line.

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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--genome-mbp", type=float, default=None, help="size of the synthetic genome (default 3100, or $BSX_BENCH_GENOME_MBP); when NOT given, $BISCUIT_HG38_INDEX / $BISCUIT_HG38_FA switch the headline record to the real genome (SURVEY 8(d) config 2)")
    ap.add_argument("--threads", type=int, default=16, help="-@ of the run: fixes the chunk size (10 Mbp x threads), like the reference")
    ap.add_argument("--host-threads", type=int, default=0, help="worker threads for the host stages; 0 = cores / ranks, capped at 64")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--single-end", action="store_true", help="every read on its own (BASELINE configs[4] shape with --read-len 1000: long single-end reads)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=100000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one chunk at a time through bsx_process_seqs (no overlap of consecutive chunks)")
    ap.add_argument("--no-long-reads", action="store_true", help="skip the third, short measurement on 1 kb single-end reads (BASELINE configs[4] shape; own subprocess)")
    ap.add_argument("--no-hard-genome", action="store_true", help="skip the second, shorter measurement on the other genome profile (run after the headline one, in a subprocess)")
    ap.add_argument("--genome-profile", choices=["clean", "hg38-like"], default="hg38-like",
                    help="hg38-like (the headline since round 5: BASELINE's metric is quoted vs hg38): i.i.d. bases + interspersed repeat families with up to a million copies, ~43 %% repeats (csrc/host/sim.c); clean: i.i.d. bases + 5 %% planted repeats (the headline of rounds 1-4, now the sub-record clean_genome)")
    ap.add_argument("--no-cli", action="store_true", help="skip the cli_end_to_end sub-record (FASTQ text in -> SAM text out through biscuit_align, tools/cli_e2e.py)")
    ap.add_argument("--sub", action="store_true", help="(internal) this run is a sub-record of another: no sub-records of its own")
    args = ap.parse_args()
    genome_given = args.genome_mbp is not None or "BSX_BENCH_GENOME_MBP" in os.environ
    if args.genome_mbp is None:
        args.genome_mbp = float(os.environ.get("BSX_BENCH_GENOME_MBP", "3100"))
    synth_mbp = args.genome_mbp   # (the sub-records stay on the synthetic genomes)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # $BSX_BENCH_SHARE_GPU=1 (a check of the N > 1 code path on a one-GPU box, never a measurement): every rank computes on GPU 0
    # and the ranks talk over gloo; the JSON line says so
    share_gpu = world > 1 and os.environ.get("BSX_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group("gloo")
        else:
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
    dev = Device(local_rank)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    # "vs hg38 index (pre-built files at $BISCUIT_HG38_INDEX; if absent, a synthetic 3.1 Gbp genome ...)".  The real
    # genome when the box has it -- $BISCUIT_HG38_INDEX = the <base> of a `biscuit index` file set (<base>.{par,dau}.{bwt,sa} +
    # <base>.bis.{ann,amb,pac}), or $BISCUIT_HG38_FA = the FASTA, indexed on the GPU at start-up -- and only for the headline record; the
    # reads are simulated from whichever genome is resident (bsx_sim_pairs reads the index's pac).
    real_genome = None
    if args.genome_profile == "hg38-like" and not args.single_end and not genome_given and not args.sub:
        real_genome = real_genome_source()
    if real_genome and real_genome[0] == "index":
        idx = Index(real_genome[1])
        dev.upload_index(idx)
    elif real_genome:
        idx = Index.from_fasta(real_genome[1])
        dev.build_index(idx, fill_host=want_cpu)
    else:
        idx = Index.synthetic(n_bases, seed=2024, n_contigs=24 if n_bases >= 1_000_000_000 else 8, profile=1 if args.genome_profile == "hg38-like" else 0)
        dev.build_index(idx, fill_host=want_cpu)
    if real_genome:
        n_bases = int(idx.l_pac)
        args.genome_mbp = n_bases / 1e6
    t_build = time.time() - t0
    barrier()

    opt = default_opt()
    opt.n_threads = threads
    # defaults: -b 0 (non-directional search: 4 strand searches per pair)
    opt.flag |= 0x10 | (0 if args.single_end else 0x2)            # MEM_F_NO_MULTI (align.c:335) | MEM_F_PE
    frag = (args.read_len, args.read_len + 400) if args.single_end else (200, 500)
    pairs_per_step = (opt.chunk_size * threads) // (2 * args.read_len)   # the reference's chunk: 10 Mbp x threads
    n_reads = pairs_per_step * 2

    L.bsx_sim_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p)]
    L.bsx_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_free_reads.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_sam_bytes.argtypes = [C.c_void_p, C.c_int64]
    L.bsx_sim_sam_bytes.restype = C.c_int64

    def gen(seed, n_pairs):
        p = C.c_void_p()
        B.check(L.bsx_sim_pairs(idx.h, n_pairs, args.read_len, seed, frag[0], frag[1], 0.005, 0.0, C.byref(p)), "sim_pairs")
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
    dev.seed_passes(reset=True)
    dev.counters(reset=True)
    dev.seed_table(reset=True)
    dev.region_work(reset=True)
    phase_tot = {}
    # the gather of the records to rank 0 (below) is part of the timed region; its connections are made here, as part of the warm-up
    import numpy as np
    from biscuit_amd.gather import ChunkGather
    gathered = [0, 0]

    def sink(k, buf):
        gathered[0] += 1
        gathered[1] += len(buf)
    gdev = torch.device("cuda", local_rank) if world > 1 and not share_gpu else torch.device("cpu")
    G = ChunkGather(rank, world, gdev, sink, max_pending=3)
    G.warm(n_reads * 600)   # a chunk's records are ~500 bytes per read: the staging buffers get their final size here too
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
    # the per-chunk alignment records (SAM text) of every rank stream to rank 0 while the following chunks are
    # aligned (biscuit_amd/gather.py: sizes, then exactly the payload, rank -> rank 0 over RCCL); rank 0 takes them in chunk
    # order and drops them (a real run writes them).  Chunk s of rank r is global chunk s * world + r.  All of it is inside
    # the timed region.  With one GPU the gather degenerates to handing the text over in this process.
    L.bsx_hook_chunk_sam.restype = C.c_int64
    L.bsx_hook_chunk_sam.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]

    def consumer():
        while True:
            k = retire_q.get()
            if k is None:
                G.close()
                return
            tr = time.time()
            nb = L.bsx_hook_chunk_sam(chunks[k], n_reads, None, 0)
            text = np.empty(nb, dtype=np.uint8)
            L.bsx_hook_chunk_sam(chunks[k], n_reads, text.ctypes.data_as(C.c_void_p), nb)
            sam_bytes_box[0] += nb
            L.bsx_sim_reset_reads(chunks[k], n_reads)
            retire_s[0] += time.time() - tr
            retired.add(k)
            G.submit((k - args.warmup) * world + rank, text)
    consumer_th = threading.Thread(target=consumer)
    consumer_th.start()
    def gather_main():
        if world > 1:
            torch.cuda.set_device(local_rank)   # the current device is per thread
        G.run()
    gather_th = threading.Thread(target=gather_main)
    gather_th.start()

    def retire(k):
        retire_q.put(k)

    def account():
        ps = B.PhaseStats()
        L.bsx_last_phase_stats(C.byref(ps))
        for f, _ in B.PhaseStats._fields_:
            phase_tot[f] = phase_tot.get(f, 0) + getattr(ps, f)
        phase_tot["_chunks"] = phase_tot.get("_chunks", 0) + 1

    loop_s = [0.0, 0.0]   # waiting for a ring slot, inside bsx_stream_push
    for s in range(args.warmup, args.warmup + args.steps):
        if args.no_pipeline:
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n_processed, n_reads, chunks[s], None), "process_seqs")
            account()
            retire(s)
        else:
            tw = time.time()
            while s - len(chunks) >= args.warmup and (s - len(chunks)) not in retired:
                time.sleep(0.001)   # the ring slot's previous use must have been consumed (it has, several steps ago)
            loop_s[0] += time.time() - tw
            tw = time.time()
            B.check(L.bsx_stream_push(stream, n_processed, n_reads, chunks[s]), "stream_push")
            loop_s[1] += time.time() - tw
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
    gather_th.join()
    torch.cuda.synchronize()
    barrier()
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    try:
        hbm_free = torch.cuda.mem_get_info(local_rank)   # (free, total) with the index, the table and every lane's buffers resident
    except Exception:
        hbm_free = (None, None)
    sam_bytes = sam_bytes_box[0]

    tmax, tot_reads = dt, n_reads * args.steps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        c = torch.tensor([tot_reads], dtype=torch.int64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(c)   # "gather" of per-GPU record counts over RCCL
        tot_reads = int(c.item())

    # ---- rooflines.  One byte definition throughout: the bytes a kernel family has to move for this input (its ALGORITHMIC bytes, stated per
    # unit in DESIGN.md section 4), over HIP-event launch times on the launch stream in the timed region.  With chunks pipelined, kernels of
    # different chunks share the device, so a launch lasts longer than it would alone; one extra chunk is therefore run unpipelined after
    # the timed region and its event times are reported next to the live ones (*_standalone).
    #   region family (K3's consumers: C1+C2+K4+C4 -- k_regions, k_regions_mid, k_x4prep/k_extl/k_ext4, k_c2r, k_regions_slab): per strand search
    #       the read (l_query B) + 32 B per SA interval + 8 B per seed occurrence; per region 56 B written + the read again and the packed
    #       reference window of its two extensions (l_query + ceil((l_query + 2 w) / 4) B)
    #   k_seedt (K1+K2): 64 B per FM block the kernel touches + a 64-byte line per entry of the table of k-mer intervals it reads.  (The
    #       reference algorithm touches four times as many blocks for the same reads: `reference_equivalent`, counted by k_seed.)
    #   k_occ (K3): 64 B per LF step of bwt_sa + 8 B per SA sample + 16 B per occurrence (rank in, position out)
    seed_p1, seed_p2, seed_ms = dev.seed_passes()   # FM blocks / table entries / HIP-event ms of each seeding pass, counted apart
    ctr = dev.counters()
    ktimes = [dev.kernel_time(k) for k in range(8)]
    rwork = dev.region_work()
    alone = None
    alone_seed_ms = None
    region_launch_ms = None
    ref_blocks_per_read, ref_seed_ms = None, None
    tab_touch = [ctr[0] + ctr[1], dev.seed_table()[0], dev.seed_table()[1]]   # FM blocks and table entries the seeding kernel read in the timed region
    if not args.no_pipeline:
        for k in range(8):
            dev.kernel_time(k, reset=True)
        dev.seed_passes(reset=True)
        extra = gen(777, pairs_per_step)
        # the region launches of this chunk one by one (the setting `tiers`: HIP events between them, printed by the library): the last HBM tier is a
        # handful of strand searches -- reads inside tandem repeats, a wavefront each -- and lasts as long as the longest of them, so
        # "regions_tiers23" of one chunk says little about the tiers that carry the load without the split
        import tempfile
        tier_txt = ""
        B.tune("tiers", "1")
        sys.stderr.flush()
        saved_fd, tf = os.dup(2), tempfile.TemporaryFile()
        os.dup2(tf.fileno(), 2)
        try:
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n_processed, n_reads, extra, None), "process_seqs(standalone)")
        finally:
            os.dup2(saved_fd, 2)
            os.close(saved_fd)
            B.tune("tiers", None)
            tf.seek(0)
            tier_txt = tf.read().decode(errors="replace")
            tf.close()
            sys.stderr.write(tier_txt)
        region_launch_ms = None
        for line in tier_txt.splitlines():
            if "region launches (ms):" in line:
                try:
                    region_launch_ms = {}
                    for part in line.split("region launches (ms):", 1)[1].split("|"):
                        part = part.strip()
                        if part:
                            name, ms = part.rsplit(" ", 1)
                            region_launch_ms[name.strip()] = float(ms)
                except ValueError:
                    region_launch_ms = None
                break
        alone = [dev.kernel_time(k) for k in range(8)]
        alone_seed_ms = dev.seed_passes()[2]   # [first pass, second pass] of the stand-alone chunk
        # The reference algorithm's FM-block touches for these reads (bwt_occ4 / bwt_2occ4 calls of bwt_smem1a and bwt_seed_strategy1:
        # deterministic integers for a given input) are counted here, after the timed region, by the same chunk through the kernel
        # that walks the FM index step by step as the reference does (k_seed.hip, the setting seed_form=classic).
        B.tune("seed_form", "classic")
        try:
            dev.counters(reset=True)
            B.check(L.bsx_process_seqs(dev.h, C.byref(opt), idx.h, n_processed, n_reads, extra, None), "process_seqs(reference block count)")
            rc_ = dev.counters()
            ref_blocks_per_read = (rc_[0] + rc_[1]) / float(n_reads)
            ref_seed_ms = dev.kernel_time(0)[0] - alone[0][0]
        finally:
            B.tune("seed_form", None)
        L.bsx_sim_free_reads(extra, n_reads)

    # Counter passes cannot run inside this process (rocprofv3 --pmc wraps a command): HBM traffic and instruction counts per launch come
    # from the committed passes of THIS command on THIS genome (profiles/*_pmc_<Mbp>mbp_<profile>.json, written by tools/profile_round.sh +
    # tools/summarize_profiles.py), and are only quoted when the workload is the profiled one.
    pmc = {}
    import glob
    prof_tag = "hg38like" if args.genome_profile == "hg38-like" else "clean"
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_%dmbp_%s.json" % (int(round(args.genome_mbp)), prof_tag))))
    if cand and args.read_len == 150 and threads == 16 and not args.single_end and not real_genome:
        with open(cand[-1]) as f:
            tj = json.load(f)
        if tj.get("_reads_per_chunk") == n_reads:   # per launch = per chunk
            pmc = tj
            pmc["_file"] = os.path.relpath(cand[-1], ROOT)
    PEAK_HBM = 8000.0       # GB/s, spec (MI355X_MICROARCH.md); a streaming copy reaches 6.3 TB/s
    GATHER_CEILING = 3500.0   # GB/s: dependent random 64-B block reads over a 3.1 GB table, four lanes per block, measured on this GPU
                              # with tools/ubench/gather64.hip (profiles/r02_gather64.txt); one lane per block: 2.7 TB/s

    # the kernel trace of this command committed under profiles/ (tools/profile_round.sh + tools/summarize_profiles.py): each family's kernel
    # durations summed per chunk, pipelined and for the stand-alone chunk -- what a reader of profiles/*_kernel_stats.csv recomputes
    trace_ms = {}
    cand_t = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_family_ms_%dmbp_%s.json" % (int(round(args.genome_mbp)), prof_tag))))
    if cand_t and args.read_len == 150 and threads == 16 and not args.single_end and not real_genome:
        with open(cand_t[-1]) as f:
            trace_ms = json.load(f)
        trace_ms["_file"] = os.path.relpath(cand_t[-1], ROOT)
    SPAN = ("HIP-event span on the chunk's launch stream inside the timed region, averaged over the steps: with %d chunks in flight the span of a launch "
            "sequence includes the time it shared the device with (or waited behind) the other chunks' kernels -- NOT exclusive kernel time; the "
            "stand-alone chunk's span (one chunk alone on the device, after the timed region) and the trace's summed kernel durations are beside it" % depth)

    def hbm_roof(name, ms, alg_bytes, extra, pmc_key=None, ms_alone=None, trace_key=None):
        # alg_bytes and ms: over the whole timed region; one launch (sequence) per step
        launches = args.steps
        if ms <= 0:
            return None
        ach = alg_bytes / (ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_HBM, "unit": "GB/s", "frac": round(ach / PEAK_HBM, 5),
             "traffic": None, "algorithmic_bytes_per_launch": alg_bytes / launches, "avg_launch_ms": round(ms / launches, 3), "avg_launch_ms_is": SPAN}
        if ms_alone:
            r["avg_launch_ms_standalone"] = round(ms_alone, 3)   # the one extra chunk
            r["achieved_standalone"] = round(alg_bytes / launches / (ms_alone * 1e-3) / 1e9, 2)
            r["frac_standalone"] = round(r["achieved_standalone"] / PEAK_HBM, 5)
        t_ = trace_ms.get(trace_key or "", None)
        if t_:
            # (rocprofv3 --kernel-trace of this command: the family's kernel durations summed, per chunk)
            r["kernel_ms_sum_from_trace"] = {"pipelined_per_chunk": t_.get("pipelined_ms_per_chunk"), "chunks": t_.get("chunks"), "by_kernel": t_.get("by_kernel"), "source": trace_ms["_file"]}
            if t_.get("pipelined_ms_per_chunk"):
                r["frac_over_trace_kernel_ms"] = round(alg_bytes / launches / (t_["pipelined_ms_per_chunk"] * 1e-3) / 1e9 / PEAK_HBM, 5)
        p = pmc.get(pmc_key or "", {})
        if p.get("FETCH_SIZE_KiB") is not None and p.get("WRITE_SIZE_KiB") is not None:
            r["traffic"] = 1024.0 * (p["FETCH_SIZE_KiB"] + p["WRITE_SIZE_KiB"])
            r["traffic_source"] = "FETCH_SIZE + WRITE_SIZE of %s (separate rocprofv3 --pmc passes of this command on this genome, one chunk: every dispatch of the family in that chunk, i.e. the same launches the bytes above are counted over)" % pmc["_file"]
        r.update(extra)
        return r

    steps_reads = float(n_reads * args.steps)
    # -- the region family: what dominates the step
    win_bytes = args.read_len + (args.read_len + 2 * opt.w + 3) // 4
    reg_bytes = rwork[4] + 32.0 * rwork[1] + 8.0 * rwork[2] + (56.0 + win_bytes) * rwork[3]
    issue = None
    p_ = pmc.get("k_regions", {})
    if p_.get("SQ_INSTS_SALU") is not None and p_.get("SQ_INSTS_VALU") is not None and alone and alone[5][1]:
        t_ = (alone[5][0] + alone[6][0]) * 1e-3
        # one scalar instruction per CU per cycle, one wave64 VALU instruction per SIMD per two cycles (256 CUs x 4 SIMDs, 2.4 GHz)
        issue = {"salu_inst_per_chunk": p_["SQ_INSTS_SALU"], "valu_inst_per_chunk": p_["SQ_INSTS_VALU"],
                 "frac_salu_issue_peak": round(p_["SQ_INSTS_SALU"] / t_ / (256 * 2.4e9), 4), "frac_valu_issue_peak": round(p_["SQ_INSTS_VALU"] * 2 / t_ / (1024 * 2.4e9), 4),
                 "over": "the stand-alone chunk's %.0f ms" % (t_ * 1e3), "counter_source": pmc["_file"]}
    roof = hbm_roof("region family (C1+C2+K4+C4): k_regions + k_regions_mid (chaining, chain filter; tables in LDS), k_x4prep/k_extl/k_ext4 (extensions ahead), k_c2r (chains -> regions), k_regions_slab x2 (HBM slabs)",
                    ktimes[5][0] + ktimes[6][0], reg_bytes,
                    {"per_read": {"strand_searches": rwork[0] / steps_reads, "sa_intervals": rwork[1] / steps_reads, "seed_occurrences": rwork[2] / steps_reads, "regions": rwork[3] / steps_reads},
                     "algorithmic_bytes_are": "per strand search the read + 32 B per SA interval + 8 B per seed occurrence; per region 56 B written + %d B (the read and the packed reference window of its extensions)" % win_bytes,
                     "issue": issue,
                     "what_bounds_it": "neither bytes nor, by the counters, instruction issue: latency and divergence -- a wavefront per strand search walks dependent LDS / HBM round trips (chaining, the chain filter), and a DP row of an extension is a dependent chain of DPP steps; see DESIGN.md section 4"},
                    "k_regions", ms_alone=(alone[5][0] + alone[6][0]) if alone and alone[5][1] else None, trace_key="region_family")
    # -- seeding: BOTH passes (the chunk-wide launch and, on a repeat-rich genome, the launch over the strand searches seeded again inside the
    # chunk's sequence), each pass's own bytes over its own time, and the two together: bytes of both over the time of both
    b1 = 64.0 * (seed_p1[0] + seed_p1[1])   # first pass: FM blocks + table lines
    b2 = 64.0 * (seed_p2[0] + seed_p2[1])   # second pass
    touched = b1 + b2
    seed_ms_both = seed_ms[0] + seed_ms[1]
    ref_eq = None
    if ref_blocks_per_read and seed_ms_both > 0:
        rb_ = 64.0 * ref_blocks_per_read * n_reads * args.steps
        ref_eq = {"fm_block_touches_per_read": ref_blocks_per_read, "bytes_per_launch": rb_ / args.steps, "rate": round(rb_ / (seed_ms_both * 1e-3) / 1e9, 2), "unit": "GB/s",
                  "frac_of_hbm_peak": round(rb_ / (seed_ms_both * 1e-3) / 1e9 / PEAK_HBM, 5),
                  "meaning": "the rate at which the kernel (both passes' time) gets through the REFERENCE algorithm's FM-block touches for these reads (bwt_occ4/bwt_2occ4 of bwt_smem1a and bwt_seed_strategy1, SURVEY 8(d); counted on one chunk by k_seed, seed_form=classic, after the timed region) -- not bytes moved: the table of k-mer intervals replaces three of four of them",
                  "kernel_without_table_ms_standalone": round(ref_seed_ms, 3) if ref_seed_ms else None}

    def pass_rec(b, ms, fm, tab, extra=None):
        if ms <= 0:
            return None
        r = {"algorithmic_bytes_per_launch": b / args.steps, "avg_launch_ms": round(ms / args.steps, 3), "achieved": round(b / (ms * 1e-3) / 1e9, 2),
             "frac": round(b / (ms * 1e-3) / 1e9 / PEAK_HBM, 5), "fm_blocks_per_read": fm / steps_reads, "table_entries_per_read": tab / steps_reads}
        r.update(extra or {})
        return r
    roof_seed = hbm_roof("k_seedt (K1+K2: SMEM seeding over the table of k-mer intervals + dependent random 64-B FM-block gathers), both passes of a chunk", seed_ms_both, touched,
                         {"kernel_touches_per_read": {"fm_blocks": (seed_p1[0] + seed_p2[0]) / steps_reads, "table_entries": (seed_p1[1] + seed_p2[1]) / steps_reads, "table_depth": tab_touch[2] if tab_touch else 0},
                          "first_pass": pass_rec(b1, seed_ms[0], seed_p1[0], seed_p1[1], {"what": "the chunk-wide launch: every strand search, a lane each"}),
                          "second_pass": pass_rec(b2, seed_ms[1], seed_p2[0], seed_p2[1], {"what": "the strand searches whose lists or budget overflowed, seeded again inside the chunk's sequence with lists eight times as long",
                                                                                           "strand_searches_per_chunk": seed_p2[3] / max(1, seed_p2[2])}),
                          "random_64B_gather_ceiling": GATHER_CEILING, "frac_of_gather_ceiling": round(touched / (seed_ms_both * 1e-3) / 1e9 / GATHER_CEILING, 4) if seed_ms_both > 0 else None,
                          "reference_equivalent": ref_eq,
                          "what_bounds_it": "vector and scalar issue of the per-lane state machine (a lane per strand search, persistent lanes, three waves per SIMD at 168 VGPRs): a trip of the wave loop is one request per lane -- an FM extension (one or two dependent random 64-B blocks, fetched by the wave as a whole) or a 16-byte table entry -- and about half its cycles are the machine that decides the next request; the second pass is a few hundred waves bound by the dependent FM steps of their longest strand searches; see DESIGN.md"},
                         "k_seed", ms_alone=(alone_seed_ms[0] + alone_seed_ms[1]) if alone_seed_ms and alone_seed_ms[0] > 0 else None, trace_key="seeding")
    roof_occ = hbm_roof("k_occ_expand + k_occ (K3: suffix-array lookups of the whole chunk)", ktimes[1][0], 64.0 * ctr[2] + 24.0 * ctr[3],
                        {"lf_steps_per_read": ctr[2] / steps_reads, "sa_lookups_per_read": ctr[3] / steps_reads}, "k_occ",
                        ms_alone=alone[1][0] if alone and alone[1][1] else None, trace_key="sa_lookup")
    # -- the whole path against the HBM peak: the same bytes, all families, over the step's wall time
    whole = None
    if roof and roof_seed and roof_occ:
        wb = (touched + 64.0 * ctr[2] + 24.0 * ctr[3] + reg_bytes) * world   # this rank's counters; ranks do the same work
        whole = {"bound": "hbm", "algorithmic_bytes_per_step": wb / args.steps, "ms_per_step": round(1e3 * tmax / args.steps, 2),
                 "achieved": round(wb / tmax / 1e9, 2), "peak": PEAK_HBM * world, "unit": "GB/s", "frac": round(wb / tmax / 1e9 / (PEAK_HBM * world), 5),
                 "bytes_are": "seeding (FM blocks and table lines touched) + K3 + the region family, as in the three rooflines above; mate rescue (K5) and CIGARs (K6) move a few hundred bytes per job"}
        if ref_blocks_per_read:
            # SURVEY 8(d)'s own per-read figure -- what the REFERENCE algorithm moves for these reads:
            #   B(read) = 64 (N_occ4 + N_2occ4_fast + N_occ) + 8 N_sa + sum over windows ceil(ref_bases / 4) + 2 l_seq + sam_bytes
            # N_occ4 + N_2occ4_fast: counted exactly (k_seed, the reference's own sequence of bwt_extend calls, one chunk after the timed region);
            # N_sa: counted exactly (one bwt_sa per seed occurrence kept); N_occ: bwt_sa's LF steps at the FILES' 1-in-32 sample = 31 per lookup in
            # expectation (a walk ends at a sampled rank with probability 1/32 per step; the device's own sample is every 2nd rank and its measured
            # mean is 1.0); windows: the two extension windows of every region (K5 / K6 windows, a few hundred bytes per job, are left out).
            n_sa = ctr[3] / steps_reads
            regions_pr = rwork[3] / steps_reads
            sam_pr = sam_bytes / steps_reads
            terms = {"fm_blocks_seeding": 64.0 * ref_blocks_per_read, "fm_blocks_bwt_sa_expected": 64.0 * 31.0 * n_sa, "sa_words": 8.0 * n_sa,
                     "reference_windows": float((win_bytes - args.read_len) * regions_pr), "read_in_and_out": 2.0 * args.read_len, "sam_text": sam_pr}
            b_read = sum(terms.values())
            whole["reference_equivalent"] = {"bytes_per_read": round(b_read, 1), "terms_bytes_per_read": {k: round(v, 1) for k, v in terms.items()},
                                             "rate": round(b_read * tot_reads / tmax / 1e9, 2), "unit": "GB/s", "frac_of_hbm_peak": round(b_read * tot_reads / tmax / 1e9 / (PEAK_HBM * world), 5),
                                             "meaning": "SURVEY 8(d)'s B(read), the bytes the reference algorithm touches for these reads, over this run's wall time: the rate at which the path gets through the reference's memory work -- not bytes this implementation moves (the table of k-mer intervals and the dense suffix-array sample replace most of them)"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(L, B, idx, dev, opt, args, ncores)

    repeats = ("with hg38-like repeat content (SINE/LINE/LTR-like families of up to a million copies, satellite arrays: ~43 % repeats)"
               if args.genome_profile == "hg38-like" else "with repeat families (5 % planted repeats of 1-5 copies)")
    if real_genome:
        repeats = "-- THE REAL GENOME: %s" % real_genome[2]
    workload = (("BASELINE configs[4] shape: 1x%d bp synthetic directional bisulfite single-end reads vs" if args.single_end else "BASELINE configs[1] shape: 2x%d bp synthetic directional bisulfite pairs vs")
                + (" a %.0f Mbp genome %s (two FM indices of %.2f G symbols each), biscuit align defaults (-b 0)" if real_genome else
                   " a SYNTHETIC %.0f Mbp genome %s "
                   "(hg38 itself is not available offline; SURVEY 8(d) config 2 fallback: an hg38-sized synthetic genome, two FM indices of %.2f G symbols each, "
                   "built on the GPU at start-up; $BISCUIT_HG38_INDEX / $BISCUIT_HG38_FA switch this record to the real genome), biscuit align defaults (-b 0)")) % (args.read_len, args.genome_mbp, repeats, 2 * n_bases / 1e9)
    if rank == 0:
        names = ["seed", "occ", "extend", "sw", "global", "regions_tier1", "regions_tiers23", "seed_again_and_host_path_batches"]   # (the last: the second seeding pass inside the chunk's sequence + K1/K2 batches of the host path)
        out = {
            "metric": "paired-end reads aligned/sec", "value": round(tot_reads / tmax, 1), "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * tmax / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": ("real genome (%s), synthetic reads" % real_genome[2]) if real_genome else "synthetic",
            "config": {"workload": workload,
                       "reads_per_step_per_gpu": n_reads, "chunk_threads(-@)": threads, "host_threads_per_gpu": host_threads, "host_cores_usable": ncores, "parallelism": ("chunk-sharded x%d" % world) + (" (CODE-PATH CHECK: all ranks on one GPU, gloo; not a measurement)" if share_gpu else ""), "chunk_pipeline_depth": depth,
                       "index_bytes_in_hbm": int(2 * (n_bases * 2 / 128 * 64 + n_bases * 2 / 2 * 8) + n_bases / 4),
                       "seed_table_bytes_in_hbm": int(2 * 16 * sum(3 ** l for l in range(1, (tab_touch[2] if tab_touch else 0) + 1)))},
            "roofline": roof,
            "roofline_seeding": roof_seed,
            "roofline_sa_lookup": roof_occ,
            "roofline_whole_path": whole,
            "record_gather": {"chunks_received_by_rank0": gathered[0], "bytes_received_by_rank0": gathered[1], "in_timed_region": True},
            "cpu_baseline": cpu,
            "kernel_ms_per_step": {names[k]: round(ktimes[k][0] / args.steps, 3) for k in range(8)},
            "kernel_ms_per_step_standalone": ({names[k]: round(alone[k][0], 3) for k in range(8)} if alone else None),
            "region_launch_ms_standalone": region_launch_ms if alone else None,   # tier 1 | tier 1b | extensions ahead | chains -> regions | tier 2 | tier 3 of the same chunk
            "strand_searches_per_step": phase_tot.get("n_tasks", 0) // max(1, phase_tot.get("_chunks", 1)), "strand_searches_chained_on_host_per_step": phase_tot.get("n_host_tasks", 0) // max(1, phase_tot.get("_chunks", 1)),
            "host_phase_s_per_chunk": {k: round(v / max(1, phase_tot.get("_chunks", 1)), 4) for k, v in phase_tot.items() if k.startswith("t_")},
            "push_loop_s_per_step": {"ring_wait": round(loop_s[0] / args.steps, 4), "in_stream_push": round(loop_s[1] / args.steps, 4)}, "sam_consumer_s_per_step": round(retire_s[0] / args.steps, 4),
            "host_cpu_s_per_step": {"user": round((ru1.ru_utime - ru0.ru_utime) / args.steps, 2), "system": round((ru1.ru_stime - ru0.ru_stime) / args.steps, 2)},
            # what a rank asks of the host at this rate (an 8-GPU node runs eight of them on its cores): CPU seconds per second of the timed region
            "host_cores_busy_per_gpu": round(((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / max(1e-9, dt), 2),
            "sam_bytes_per_read": round(sam_bytes / float(n_reads * args.steps), 1),
            "genome_and_index_build_s": round(t_build, 1), "device": dev.name,
            "hbm_bytes_free_of_total_after_the_run": list(hbm_free),
        }
    if not args.no_pipeline:
        L.bsx_stream_close(stream)
    for c in list.__iter__(chunks):
        L.bsx_sim_free_reads(c, n_reads)
    if rank == 0:
        # Sub-records, each in a process of its own after this one has given the device back (the headline line must not depend on them):
        #   clean_genome     the same measurement, shorter, on the genome of rounds 1-4's headline (i.i.d. bases + 5 % planted repeats)
        #   long_reads       BASELINE configs[4] shape (1 kb single-end reads), with its own roofline and CPU baseline
        #   cli_end_to_end   SURVEY 8(d)'s second number: FASTQ text in -> SAM text out through the command line (tools/cli_e2e.py)
        if world == 1 and not args.sub and not args.no_pipeline:
            dev.close()
            idx.close()
            keep = ("value", "unit", "steps", "warmup", "ms_per_step", "roofline", "roofline_seeding", "roofline_whole_path", "cpu_baseline", "kernel_ms_per_step",
                    "kernel_ms_per_step_standalone", "region_launch_ms_standalone", "strand_searches_chained_on_host_per_step", "host_phase_s_per_chunk", "host_cpu_s_per_step",
                    "push_loop_s_per_step", "host_cores_busy_per_gpu")

            def sub_run(key, extra_args, timeout):
                cmd = [sys.executable, os.path.abspath(__file__), "--sub", "--genome-mbp", str(synth_mbp), "--threads", str(threads)] + extra_args
                if args.no_cpu_baseline:
                    cmd.append("--no-cpu-baseline")
                try:
                    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
                    sub = json.loads(pr.stdout.decode().strip().split("\n")[-1])
                    out[key] = {k: sub.get(k) for k in keep}
                    out[key]["workload"] = sub["config"]["workload"]
                except Exception as e:
                    out[key] = {"error": repr(e)[:300]}
            if not args.no_hard_genome:
                other = "clean" if args.genome_profile == "hg38-like" else "hg38-like"
                sub_run("clean_genome" if other == "clean" else "hg38_like_genome",
                        ["--genome-profile", other, "--read-len", str(args.read_len), "--steps", str(max(2, min(8 if other == "clean" else 4, args.steps))), "--warmup", "2",
                         "--cpu-sample-pairs", str(max(1000, args.cpu_sample_pairs // 5))], 1500)
            if not args.no_long_reads:
                # (a chunk is 160 Mbp of reads: 160 k of them; the CPU baseline's sample is 6 000 reads)
                sub_run("long_reads", ["--genome-profile", "clean", "--single-end", "--read-len", "1000", "--steps", "6", "--warmup", "2", "--cpu-sample-pairs", "3000"], 1200)   # (three steps were a fill and a drain: 61-75 k reads/s from box to box)
            if not args.no_cli:
                try:
                    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_e2e.py"), "--genome-mbp", str(synth_mbp), "--profile", "1" if args.genome_profile == "hg38-like" else "0",
                                         "--threads", str(threads), "--chunks", "7", "--out", "/dev/null", "--json"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
                    out["cli_end_to_end"] = json.loads(pr.stdout.decode().strip().split("\n")[-1])
                except Exception as e:
                    out["cli_end_to_end"] = {"error": repr(e)[:300]}
        # the driver's record keeps the top level of this line: the sub-records' headline values are repeated there
        for key, top in (("clean_genome", "clean_genome_reads_s"), ("long_reads", "long_reads_s"), ("cli_end_to_end", "cli_reads_s"), ("hg38_like_genome", "hg38_like_genome_reads_s")):
            if isinstance(out.get(key), dict) and out[key].get("value") is not None:
                out[top] = out[key]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def real_genome_source():
    """the hg38 BISCUIT index at $BISCUIT_HG38_INDEX (the <base> of the seven files), else the FASTA at $BISCUIT_HG38_FA
    (indexed on the GPU at start-up); None when neither is usable -- the synthetic hg38-sized genome is the stated fallback."""
    base = os.environ.get("BISCUIT_HG38_INDEX")
    if base:
        need = [base + e for e in (".par.bwt", ".par.sa", ".dau.bwt", ".dau.sa", ".bis.ann", ".bis.amb", ".bis.pac")]
        if all(os.path.exists(f) for f in need):
            return ("index", base, "index files %s.*" % base)
        sys.stderr.write("[bench] $BISCUIT_HG38_INDEX=%s: not a complete index file set (%s missing); falling back\n" % (base, ", ".join(os.path.basename(f) for f in need if not os.path.exists(f))))
    fa = os.environ.get("BISCUIT_HG38_FA")
    if fa:
        if os.path.exists(fa):
            return ("fasta", fa, "FASTA %s, indexed on the GPU at start-up" % fa)
        sys.stderr.write("[bench] $BISCUIT_HG38_FA=%s does not exist; falling back\n" % fa)
    return None


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


def cpu_baseline(L, B, idx, dev, opt, args, ncores):
    """The CPU path on a bounded sample of the same workload, all usable host cores: this repository's C host pipeline (the reference's
    memchain.c / mem_alnreg.c / mem_pair.c / mem_alnreg_format.c cannot be built offline: un-vendored headers) over
    the REFERENCE'S OWN kernels -- oracle/_ref/libbiscuit_ref.so = lib/aln/bwt.c and ksw.c (SSE2 ksw_u8/ksw_i16 included) compiled where
    they lie -- which is where a CPU run spends its time.  The same sample through the plain scalar restatement of the kernels
    (oracle/port.c), 2.5-4x slower, is reported next to it; round 2 quoted only that one."""
    import oracle_lib
    o = B.Opt.from_buffer_copy(opt)
    os.environ["BSX_HOST_THREADS"] = str(ncores)
    n_pairs = args.cpu_sample_pairs
    L.bsx_process_seqs_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.bsx_sim_reset_reads.argtypes = [C.c_void_p, C.c_int64]

    import zlib

    def sam_crc(p, n):   # the records' SAM text, read by read
        reads = C.cast(p, C.POINTER(B.Read))
        crc, nb = 0, 0
        for i in range(n):
            t = C.string_at(reads[i].sam) if reads[i].sam else b""
            crc = zlib.crc32(t, crc)
            nb += len(t)
        return crc, nb

    def run(port, pairs, check_hip=False):
        p = C.c_void_p()
        B.check(L.bsx_sim_pairs(idx.h, pairs, args.read_len, 999, args.read_len if args.single_end else 200, args.read_len + 400 if args.single_end else 500, 0.005, 0.0, C.byref(p)), "sim_pairs")
        be = port.backend()
        t0 = time.time()
        B.check(L.bsx_process_seqs_backend(C.byref(be), C.byref(o), idx.h, 0, pairs * 2, p, None), "cpu baseline")
        dt = time.time() - t0
        same = None
        if check_hip:   # the same reads through the HIP path: the two SAM texts must be the same bytes
            want = sam_crc(p, pairs * 2)
            L.bsx_sim_reset_reads(p, pairs * 2)
            B.check(L.bsx_process_seqs(dev.h, C.byref(o), idx.h, 0, pairs * 2, p, None), "process_seqs(the CPU sample through HIP)")
            got = sam_crc(p, pairs * 2)
            same = {"sam_identical": want == got, "sam_crc32_cpu": "%08x" % want[0], "sam_crc32_hip": "%08x" % got[0], "sam_bytes": want[1]}
        L.bsx_sim_free_reads(p, pairs * 2)
        return pairs * 2 / dt, dt, same

    out = None
    ref = oracle_lib.Port(idx, n_threads=ncores)
    if ref.use_reference_kernels():
        v, dt, same = run(ref, n_pairs, check_hip=True)
        out = {"value": round(v, 1), "unit": "reads/s", "cores": ncores, "kind": "port",
               "kernels": "the reference's own (oracle/_ref: lib/aln/bwt.c, ksw.c compiled where they lie) under this repository's C host pipeline",
               "sample": "%d %s of the same workload, one chunk, %.1f s" % (n_pairs * (2 if args.single_end else 1), "reads" if args.single_end else "pairs", dt)}
        out.update(same)
    small = max(1000, n_pairs // 4) if out else n_pairs
    v2, dt2, same2 = run(oracle_lib.Port(idx, n_threads=ncores), small, check_hip=out is None)
    if out is None:
        out = {"value": round(v2, 1), "unit": "reads/s", "cores": ncores, "kind": "port", "kernels": "scalar C restatement (oracle/port.c); oracle/_ref is absent",
               "sample": "%d pairs of the same workload, one chunk, %.1f s" % (small, dt2)}
        out.update(same2)
    else:
        out["scalar_restatement_kernels"] = {"value": round(v2, 1), "sample": "%d pairs, %.1f s" % (small, dt2),
                                             "slowdown_vs_reference_kernels": round(out["value"] / v2, 2)}
    return out


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        # a failure in the middle of the stream leaves worker threads behind (the pipeline's front halves, the SAM consumer):
        # report and leave at once instead of waiting for them
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
