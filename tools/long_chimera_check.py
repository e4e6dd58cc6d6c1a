#!/usr/bin/env python3
"""Long chimeric reads in the default mode: 60 reads glued from 20 / 150 / 400 pieces of different genes (12 ... 220 kb, hundreds of
chains per read: the per-read capacities of chain_kernel grow with the read length) through the reference and examples/kmahip_map -chain.
usage (GPU box): python3 tools/long_chimera_check.py"""
import gzip, os, subprocess, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import numpy as np
from kma_amd import synth
ROOT = "/root/repo"
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(5)
names, seqs = synth.make_gene_db(200, 4, 800, 1500, 0.04, seed=77)
prefix = os.path.join(tmp, "db")
synth.write_fasta(prefix + ".fsa", names, seqs)
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
for pieces in ([int(x) for x in os.environ["PIECES"].split(",")] if os.environ.get("PIECES") else (20, 150, 400)):
    reads = []
    for i in range(60):
        parts = []
        for _ in range(pieces):
            s = seqs[int(rng.integers(0, len(seqs)))]
            L = int(rng.integers(300, 700)); a = int(rng.integers(0, len(s) - L))
            r = s[a:a + L].copy()
            if rng.random() < 0.5: r = synth.revcomp_codes(r)
            x = rng.random(len(r)) < 0.03
            r[x] = (r[x] + rng.integers(1, 4, int(x.sum()), dtype=np.uint8)) & 3
            parts.append(r)
            parts.append(rng.integers(0, 4, int(rng.integers(0, 80)), dtype=np.uint8))
        reads.append(np.concatenate(parts))
    fq = os.path.join(tmp, "r.fq")
    synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    t0 = time.perf_counter()
    subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t1 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-chain"], stderr=subprocess.PIPE)
    t2 = time.perf_counter()
    if r.returncode:
        print(pieces, "pieces: kmahip_map failed:", r.stderr.decode().strip().splitlines()[-1:], flush=True); continue
    same = [open(os.path.join(tmp, f"ref.{e}"), "rb").read() == open(os.path.join(tmp, f"got.{e}"), "rb").read() for e in ("res", "fsa")]
    same.append(gzip.open(os.path.join(tmp, "ref.frag.gz")).read() == gzip.open(os.path.join(tmp, "got.frag.gz")).read())
    if not same[0]:
        a = open(os.path.join(tmp, "ref.res")).read().splitlines(); b = open(os.path.join(tmp, "got.res")).read().splitlines()
        print("  .res lines", len(a), len(b))
        for x, y in list(zip(a, b)):
            if x != y:
                print("  ref:", x); print("  got:", y)
    print(pieces, "pieces per read (", max(len(x) for x in reads), "nt ): reference", round(t1 - t0, 2), "s, kmahip_map", round(t2 - t1, 2), "s; identical", same, "rows", gzip.open(os.path.join(tmp, "ref.frag.gz")).read().count(b"\n"), flush=True)
