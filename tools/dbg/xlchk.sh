set -e
cd /root/repo
python - <<'PY'
import sys, os, subprocess
sys.path.insert(0, "tests")
import simdata
from biscuit_amd.api import Index
d = "/tmp/xlchk"; os.makedirs(d, exist_ok=True)
contigs = simdata.make_genome(1000000, seed=21, n_contigs=3)
simdata.write_genome(d + "/g.fa", contigs)
Index.build(d + "/g.fa", d + "/g").close()
p100 = simdata.make_pairs(contigs, 5000, 100, 1, frag=(180, 320), sub=0.005)
simdata.write_fastq(d + "/a1.fq", [(n, a) for n, a, b in p100])
simdata.write_fastq(d + "/a2.fq", [(n, b) for n, a, b in p100])
e = dict(os.environ); e["BSX_XL_CHECK"] = "1"
p = subprocess.run(["/root/repo/biscuit_amd/biscuit_align", "-@", "4", "g", "a1.fq", "a2.fq"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
print(p.stderr.decode()[-6000:])
PY
