import os, sys, subprocess
sys.path.insert(0, "tests")
import simdata
ROOT = os.getcwd()
HIP = os.path.join(ROOT, "biscuit_amd", "biscuit_align")
d = "/tmp/longdbg"; os.makedirs(d, exist_ok=True)
from biscuit_amd.api import Index
contigs = simdata.make_genome(1000000, seed=21, n_contigs=3)
simdata.write_genome(d + "/g.fa", contigs)
Index.build(d + "/g.fa", d + "/g").close()
simdata.write_fastq(d + "/long.fq", simdata.make_single(contigs, 300, 1000, 5))
def run(env):
    e = dict(os.environ); e.update(env)
    p = subprocess.run([HIP, "-@", "4", "g", "long.fq"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return [l for l in p.stdout.decode().split("\n") if not l.startswith("@")], p.stderr.decode()
a, ea = run({"BSX_PHASES": "1"})
b, eb = run({"BSX_TUNE": "host_chain=1"})
print([l for l in ea.split("\n") if "M::regions]" in l][:3])
nd = 0
for x, y in zip(a, b):
    if x != y:
        fx, fy = x.split("\t"), y.split("\t")
        print(fx[0], [(u, v) for u, v in zip(fx, fy) if u != v][:6], len(fx), len(fy))
        nd += 1
        if nd > 12: break
print("lines", len(a), len(b), "diff", sum(1 for x, y in zip(a, b) if x != y))
