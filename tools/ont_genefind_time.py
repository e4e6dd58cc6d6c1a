#!/usr/bin/env python3
"""Gene finding in long reads, the default mode's typical job: ONT-like reads of ~10 kb, each carrying one or two genes of the 5 k-gene
database inside unrelated sequence (10 % errors), `kma -i ont.fq -t_db db -bcNano` against examples/kmahip_map -chain -bcNano.
usage (GPU box): python3 tools/ont_genefind_time.py [reads]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp(prefix="ontgf_")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
rng = np.random.default_rng(21)
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
synth.write_fasta(prefix + ".fsa", names, seqs)
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
reads = []
for i in range(n):
    parts = [rng.integers(0, 4, int(rng.integers(1000, 6000)), dtype=np.uint8)]
    for _ in range(int(rng.integers(1, 3))):
        g = seqs[int(rng.integers(0, len(seqs)))]
        parts.append(synth.revcomp_codes(g) if rng.random() < 0.5 else g)
        parts.append(rng.integers(0, 4, int(rng.integers(1000, 5000)), dtype=np.uint8))
    r = np.concatenate(parts)
    reads.append(synth.make_long_reads(r, 1, read_len=len(r), seed=100 + i)[0])
fq = os.path.join(tmp, "ont.fq")
synth.write_fastq(fq, reads, prefix="r", qual=b"5")
t0 = time.perf_counter()
subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-bcNano", "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t1 = time.perf_counter()
env = dict(os.environ)
r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-chain", "-bcNano"], stderr=subprocess.PIPE, env=env)
t2 = time.perf_counter()
if r.returncode:
    sys.exit("kmahip_map failed: " + r.stderr.decode()[-300:])
same = [open(os.path.join(tmp, f"ref.{e}"), "rb").read() == open(os.path.join(tmp, f"got.{e}"), "rb").read() for e in ("res", "fsa")]
same.append(gzip.open(os.path.join(tmp, "ref.frag.gz")).read() == gzip.open(os.path.join(tmp, "got.frag.gz")).read())
print(f"{n} reads, {sum(len(x) for x in reads) / 1e6:.1f} Mbases: reference -t 1 {t1 - t0:.1f} s, kmahip_map {t2 - t1:.2f} s; identical {same}; rows "
      f"{gzip.open(os.path.join(tmp, 'ref.frag.gz')).read().count(bytes([10]))}", flush=True)
print(r.stderr.decode().strip().splitlines()[-1][:400] if not os.environ.get("KMAHIP_DEBUG_TIMING") else r.stderr.decode(), flush=True)
