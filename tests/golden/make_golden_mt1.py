#!/usr/bin/env python3
"""Fixture of BASELINE config C4 in small: `kma -i reads.fq -o out -t_db db -Mt1 1 -bcNano -t 1 -sam` (runKMA_Mt1, mt1.c:86-500) on
ONT-like reads against ONE template. Raw reads go straight to stage 3c: anker_rc picks the strand (align.c:780-991), KMA()
chains and joins with traceback (hundreds of small NW problems and a few banded ones per long read), alnToMat piles up in
stream order, callConsensus calls with nanoCaller / significantAnd90Nuc (-bcNano, kma.c:762-766).

Inputs are synthetic and seeded here; the outputs are the reference's own: out.res, out.fsa.gz (consensus), out.frag.gz
(fragment rows in assembly order) and out.sam.tsv.gz (qname, flag, rname, pos, mapq, cigar, AS of every record -- unmapped
records keep cigar '*'). Only data is stored."""
import gzip
import lzma
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
OUT = os.path.join(HERE, "mt1")


def make_inputs():
    from kma_amd import synth
    rng = np.random.default_rng(4)
    G = 60000
    genome = rng.integers(0, 4, size=G, dtype=np.uint8)
    # a tandem repeat and an inverted repeat: duplicated k-mers (the several-MEMs-per-lookup branch of anker_rc)
    genome[20000:20300] = genome[19700:20000]
    genome[41000:41200] = synth.revcomp_codes(genome[40000:40200])
    reads = synth.make_long_reads(genome, 80, read_len=3000, sub=0.04, dele=0.03, ins=0.03, seed=5)
    reads += synth.make_long_reads(genome, 14, read_len=9000, sub=0.04, dele=0.03, ins=0.03, seed=6)
    reads += synth.make_long_reads(genome, 40, read_len=120, sub=0.05, dele=0.03, ins=0.03, seed=7)     # preseed decides for short reads
    reads += synth.make_long_reads(genome, 20, read_len=40, sub=0.02, dele=0.0, ins=0.0, seed=8)
    for i in range(8):                      # a foreign chunk inside: joins beyond the band on both sides (NW_band)
        r = synth.make_long_reads(genome, 1, read_len=2500, sub=0.03, dele=0.02, ins=0.02, seed=100 + i)[0]
        p, L = int(rng.integers(500, 1500)), int(rng.integers(70, 260))
        reads.append(np.concatenate([r[:p], rng.integers(0, 4, size=L, dtype=np.uint8), r[p:]]))
    for i in range(6):                      # N's
        r = synth.make_long_reads(genome, 1, read_len=2000, sub=0.03, dele=0.02, ins=0.02, seed=200 + i)[0].copy()
        r[rng.integers(0, len(r), size=int(rng.integers(1, 8)))] = 4
        reads.append(r)
    for i in range(4):                      # unmappable
        reads.append(rng.integers(0, 4, size=1500, dtype=np.uint8))
    for i in range(6):                      # exact copies, both strands
        st = int(rng.integers(0, G - 800))
        r = genome[st:st + 800].copy()
        reads.append(np.ascontiguousarray(synth.revcomp_codes(r) if i & 1 else r))
    reads.append(np.concatenate([rng.integers(0, 4, size=150, dtype=np.uint8), genome[:1200]]))      # overhangs at either template end
    reads.append(np.concatenate([genome[-1500:], rng.integers(0, 4, size=200, dtype=np.uint8)]))
    reads.append(np.ascontiguousarray(synth.revcomp_codes(genome[-700:])))
    reads.append(genome[:900].copy())
    reads = [reads[i] for i in rng.permutation(len(reads))]
    return genome, reads


def main():
    from kma_amd import synth
    if not os.path.exists(KMA):
        sys.exit("oracle/_ref/kma missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    genome, reads = make_inputs()
    with tempfile.TemporaryDirectory() as tmp:
        fsa, fq, prefix = os.path.join(tmp, "g.fsa"), os.path.join(tmp, "reads.fq"), os.path.join(tmp, "db")
        synth.write_fasta(fsa, ["genome60k"], [genome])
        synth.write_fastq(fq, reads, prefix="ont", qual=b"5")
        subprocess.run([KMA, "index", "-i", fsa, "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        sam = subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "out"), "-t_db", prefix, "-Mt1", "1", "-bcNano", "-t", "1", "-sam"],
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        rows = []
        for line in sam.splitlines():
            if line.startswith("@"):
                continue
            c = line.split("\t")
            AS = ([x for x in c[11:] if x.startswith("AS:i:")] or ["AS:i:0"])[0][5:]
            rows.append("\t".join([c[0], c[1], c[2], c[3], c[4], c[5], AS]))

        def gz(src, dst):
            with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
                g.write(f.read())
        with gzip.GzipFile(os.path.join(OUT, "out.sam.tsv.gz"), "wb", mtime=0) as g:
            g.write(("\n".join(rows) + "\n").encode())
        gz(os.path.join(tmp, "out.fsa"), os.path.join(OUT, "out.fsa.gz"))
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(OUT, "out.res"))
        with gzip.open(os.path.join(tmp, "out.frag.gz"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "out.frag.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        gz(fq, os.path.join(OUT, "reads.fq.gz"))
        gz(fsa, os.path.join(OUT, "db.fsa.gz"))
        with open(prefix + ".comp.b", "rb") as f, lzma.open(os.path.join(OUT, "db.comp.b.xz"), "wb", preset=9) as g:
            shutil.copyfileobj(f, g)
        for ext in (".length.b", ".seq.b", ".name"):
            shutil.copy(prefix + ext, os.path.join(OUT, "db" + ext))
        print(len(reads), "reads,", sum(len(r) for r in reads), "bases;", sum(1 for r in rows if r.split("\t")[5] != "*"), "mapped SAM records")


if __name__ == "__main__":
    main()
