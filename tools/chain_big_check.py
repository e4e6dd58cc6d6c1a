import gzip, os, subprocess, sys, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from kma_amd import synth
from test_oracle_golden import _chimeric_reads
ROOT = "/root/repo"
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(4242)
names, seqs = synth.make_gene_db(200, 10, 400, 1500, 0.045, seed=999)
prefix = os.path.join(tmp, "db")
synth.write_fasta(prefix + ".fsa", names, seqs)
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
reads = _chimeric_reads(seqs, 100000, rng, with_n=True)
# keep N's away from the first 16 bases (the reference reads beyond its buffer there)
for r in reads:
    r[:16][r[:16] == 4] = 0
fq = os.path.join(tmp, "r.fq")
synth.write_fastq(fq, reads)
for mf in ([], ["-mf", "5000"]):
    subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-t", "1"] + mf, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-chain"] + mf, check=True, stderr=subprocess.DEVNULL)
    a = [open(os.path.join(tmp, f"{x}.res"), "rb").read() for x in ("ref", "got")]
    b = [open(os.path.join(tmp, f"{x}.fsa"), "rb").read() for x in ("ref", "got")]
    c = [gzip.open(os.path.join(tmp, f"{x}.frag.gz")).read() for x in ("ref", "got")]
    print(mf, "res", a[0] == a[1], a[0].count(b"\n"), "fsa", b[0] == b[1], "frag", c[0] == c[1], c[0].count(b"\n"), flush=True)
