"""-m gpu: the N > 1 path of bench.py (one process per rank, barrier-bracketed timing, MAX over ranks, the per-chunk record
gather to rank 0 inside the timed region) run with two ranks.  A test box has one GPU: both ranks compute on it and talk over
gloo ($BSX_BENCH_SHARE_GPU, labelled in the JSON line); with one GPU per rank the same code runs over RCCL."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line():
    env = dict(os.environ, BSX_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--genome-mbp", "8", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines          # rank 0 prints, once
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert "CODE-PATH CHECK" in d["config"]["parallelism"]
    g = d["record_gather"]
    assert g["in_timed_region"] and g["chunks_received_by_rank0"] == 6 and g["bytes_received_by_rank0"] > 6 * 1000000
    # whole-job rate: both ranks' reads over the slowest rank's time
    assert abs(d["value"] - 2 * d["config"]["reads_per_step_per_gpu"] * 3 / (d["ms_per_step"] * 3e-3)) < 1e-3 * d["value"]


def test_single_rank_bench_line_small():
    """the default command's line at a small genome: the contract keys, the rooflines' byte definitions, and the CPU baseline's sample
    through both paths with identical SAM (bench.py, cpu_baseline.sam_identical)"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--genome-mbp", "8", "--cpu-sample-pairs", "3000",
           "--no-long-reads", "--no-cli", "--no-hard-genome"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "hg38-like" in d["config"]["workload"] and d["n_gpus"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    assert r["per_read"]["regions"] > 0.5 and r["per_read"]["strand_searches"] == 2.0
    s = d["roofline_seeding"]
    assert 0 < s["frac"] < 1 and s["reference_equivalent"]["fm_block_touches_per_read"] > s["kernel_touches_per_read"]["fm_blocks"]
    w = d["roofline_whole_path"]
    assert w["algorithmic_bytes_per_step"] >= r["algorithmic_bytes_per_launch"] + s["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["sam_identical"] is True and c["sam_crc32_cpu"] == c["sam_crc32_hip"] and c["cores"] >= 1 and c["value"] > 0


def test_real_genome_hook(tmp_path):
    """"vs hg38 index (pre-built files at $BISCUIT_HG38_INDEX; if absent, a synthetic 3.1 Gbp genome ...)".  bench.py takes
    the real genome when the box has it -- index files at $BISCUIT_HG38_INDEX, or the FASTA at $BISCUIT_HG38_FA indexed on the GPU -- and says so
    in `data` and `config.workload`; with neither (or an incomplete file set) it falls back and says that.  A small genome stands in for hg38."""
    import simdata
    from biscuit_amd.api import Index
    d = str(tmp_path)
    simdata.write_genome(d + "/g.fa", simdata.make_genome(3000000, seed=31, n_contigs=3))
    Index.build(d + "/g.fa", d + "/g").close()
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--threads", "1", "--cpu-sample-pairs", "1000",
            "--no-long-reads", "--no-cli", "--no-hard-genome"]
    env = {k: v for k, v in os.environ.items() if k not in ("BISCUIT_HG38_INDEX", "BISCUIT_HG38_FA", "BSX_BENCH_GENOME_MBP")}
    for how, e in (("index", {"BISCUIT_HG38_INDEX": d + "/g"}), ("fasta", {"BISCUIT_HG38_FA": d + "/g.fa"})):
        p = subprocess.run(base, cwd=ROOT, env=dict(env, **e), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
        assert r["data"].startswith("real genome") and "synthetic reads" in r["data"], r["data"]
        assert "THE REAL GENOME" in r["config"]["workload"] and "SYNTHETIC" not in r["config"]["workload"]
        assert r["value"] > 0 and r["cpu_baseline"]["sam_identical"] is True
        assert abs(r["config"]["index_bytes_in_hbm"] / (2 * (3e6 * 2 / 128 * 64 + 3e6 * 2 / 2 * 8) + 3e6 / 4) - 1) < 0.01   # the 3 Mbp genome is what is resident
    # an incomplete file set is reported and the synthetic genome takes over (a --genome-mbp given on the command line never looks at the hook)
    os.remove(d + "/g.dau.sa")
    p = subprocess.run(base + ["--no-cpu-baseline", "--genome-mbp", "8"], cwd=ROOT, env=dict(env, BISCUIT_HG38_INDEX=d + "/g"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert r["data"] == "synthetic" and "SYNTHETIC 8 Mbp" in r["config"]["workload"]
