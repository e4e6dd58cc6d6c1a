#!/usr/bin/env python3
"""Add the stage-3c per-read tap to the fixtures: the reference's SAM records (`kma ... -sam`, sam.c:114-204), which carry
what KMA()'s traceback produced for every read that stage 3c kept: POS, the extended CIGAR (= X I D S) and AS (read score
incl. the end bonus), plus the FLAG ConClave left. Runs oracle/_ref/kma on the inputs already committed next to this script
(index files + FASTQ) and stores the columns qname, flag, rname, pos, mapq, cigar, AS of the mapped records as
out.sam.tsv.gz, and the run's consensus FASTA as out.fsa.gz. Only data is stored."""
import gzip
import lzma
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")


def unpack(src, tmp):
    prefix = os.path.join(tmp, "db")
    with lzma.open(os.path.join(src, "db.comp.b.xz"), "rb") as f, open(prefix + ".comp.b", "wb") as g:
        shutil.copyfileobj(f, g)
    for ext in (".length.b", ".seq.b", ".name"):
        shutil.copy(os.path.join(src, "db" + ext), prefix + ext)
    fq = []
    for name in ("reads.fq.gz", "r1.fq.gz", "r2.fq.gz"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            out = os.path.join(tmp, name[:-3])
            with gzip.open(p, "rb") as f, open(out, "wb") as g:
                shutil.copyfileobj(f, g)
            fq.append(out)
    return prefix, fq


def make(name):
    src = os.path.join(HERE, name)
    with tempfile.TemporaryDirectory() as tmp:
        prefix, fq = unpack(src, tmp)
        inp = ["-i", fq[0]] if len(fq) == 1 else ["-ipe", fq[0], fq[1], "-apm", "p"]
        cmd = [KMA] + inp + ["-o", os.path.join(tmp, "out"), "-t_db", prefix, "-1t1", "-t", "1", "-sam"]
        sam = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        rows = []
        for line in sam.splitlines():
            if line.startswith("@"):
                continue
            c = line.split("\t")
            if c[5] == "*":
                continue
            AS = [x for x in c[11:] if x.startswith("AS:i:")][0][5:]
            rows.append("\t".join([c[0], c[1], c[2], c[3], c[4], c[5], AS]))
        with gzip.GzipFile(os.path.join(src, "out.sam.tsv.gz"), "wb", mtime=0) as g:
            g.write(("\n".join(rows) + "\n").encode())
        # the consensus sequences of the same run (printConsensus, printconsensus.c:24-61)
        with open(os.path.join(tmp, "out.fsa"), "rb") as f, gzip.GzipFile(os.path.join(src, "out.fsa.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        print(name, len(rows), "mapped SAM records")


if __name__ == "__main__":
    if not os.path.exists(KMA):
        sys.exit("oracle/_ref/kma missing: run `make -C oracle ref` first")
    for n in ("se", "long", "pe"):
        make(n)
