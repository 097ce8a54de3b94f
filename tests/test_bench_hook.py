"""CPU: bench.py's choice of genome: $BISCUIT_HG38_INDEX -> $BISCUIT_HG38_FA -> the synthetic fallback."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_real_genome_source(tmp_path, monkeypatch, capsys):
    b = _bench()
    monkeypatch.delenv("BISCUIT_HG38_INDEX", raising=False)
    monkeypatch.delenv("BISCUIT_HG38_FA", raising=False)
    assert b.real_genome_source() is None
    base = str(tmp_path / "hg38.fa")
    for e in (".par.bwt", ".par.sa", ".dau.bwt", ".dau.sa", ".bis.ann", ".bis.amb"):
        open(base + e, "wb").close()
    monkeypatch.setenv("BISCUIT_HG38_INDEX", base)
    assert b.real_genome_source() is None and ".bis.pac missing" in capsys.readouterr().err      # incomplete: said so, fallback
    open(base + ".bis.pac", "wb").close()
    kind, path, label = b.real_genome_source()
    assert kind == "index" and path == base and base in label
    monkeypatch.delenv("BISCUIT_HG38_INDEX")
    monkeypatch.setenv("BISCUIT_HG38_FA", str(tmp_path / "nope.fa"))
    assert b.real_genome_source() is None and "does not exist" in capsys.readouterr().err
    open(str(tmp_path / "hg38.fa"), "w").write(">c\nACGT\n")
    monkeypatch.setenv("BISCUIT_HG38_FA", str(tmp_path / "hg38.fa"))
    assert b.real_genome_source()[0] == "fasta"
    monkeypatch.setenv("BISCUIT_HG38_INDEX", base)     # the index files win over the FASTA
    assert b.real_genome_source()[0] == "index"
