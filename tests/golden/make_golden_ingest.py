#!/usr/bin/env python3
"""Fixtures for the stage-1 ingest (kmahip_ingest_*): awkward FASTQ / FASTA inputs and the S1 stream the compiled
reference (oracle/_ref/kma ... -s1) writes for them under several trimming settings.

    python3 tests/golden/make_golden_ingest.py        (needs /root/reference -> `make -C oracle ref`)

Inputs (seeded, small): qualities with bad ends and bad stretches, N / lower case / IUPAC codes, DOS line ends, a
phred-64 file, a FASTA file with wrapped lines and N ends, a mate pair of files. Expected outputs: tests/golden/ingest/
<case>.<setting>.s1.gz; the inputs are committed next to them."""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from kma_amd import formats, synth  # noqa: E402

KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
OUT = os.path.join(ROOT, "tests", "golden", "ingest")
SETTINGS = {                       # name -> extra reference flags
    "default": [],
    "mp25": ["-mp", "25"],
    "eq15": ["-eq", "15"],
    "eq25ml40": ["-eq", "25", "-ml", "40"],
    "mi12": ["-mi", "12"],
    "ml60xl130": ["-ml", "60", "-xl", "130"],
}
ALPHA = b"ACGT"


def rand_read(rng, L):
    s = bytearray(ALPHA[i] for i in rng.integers(0, 4, L))
    for _ in range(int(rng.integers(0, 4))):                       # a few N / IUPAC / lower-case bases
        p = int(rng.integers(0, L))
        s[p] = rng.choice(list(b"NnRYSWKMBDHVXacgtrykm"))
    if rng.random() < 0.1:
        k = min(L, int(rng.integers(1, 12)))
        s[:k] = b"N" * k
    if rng.random() < 0.1:
        k = min(L, int(rng.integers(1, 12)))
        s[-k:] = b"N" * k
    return bytes(s)


def rand_qual(rng, L, base=33):
    q = rng.integers(25, 41, L)
    if rng.random() < 0.6:
        q[: int(rng.integers(1, 30))] = rng.integers(2, 24, 1)
    if rng.random() < 0.6:
        q[-int(rng.integers(1, 40)):] = rng.integers(2, 22, 1)
    for _ in range(int(rng.integers(0, 3))):
        a = int(rng.integers(0, L))
        q[a:a + int(rng.integers(1, 25))] = rng.integers(2, 20, 1)
    if rng.random() < 0.05:
        q[:] = rng.integers(2, 15, L)
    return bytes((q + base).astype(np.uint8))


def write_fq(path, n, seed, base=33, eol=b"\n", lens=(30, 160)):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(*lens))
            f.write(b"@q%d some comment \t" % i + eol + rand_read(rng, L) + eol + b"+" + eol + rand_qual(rng, L, base) + eol)


def write_fa(path, n, seed):
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(20, 400))
            s = rand_read(rng, L)
            f.write(b">f%d descr\n" % i)
            for a in range(0, L, 60):
                f.write(s[a:a + 60] + b"\n")


def main():
    if not os.path.exists(KMA):
        sys.exit("build the reference first: make -C oracle ref")
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    with tempfile.TemporaryDirectory() as tmp:
        names, seqs = synth.make_gene_db(3, 2, 300, 400, 0.02, seed=5)
        fa = os.path.join(tmp, "db.fsa")
        synth.write_fasta(fa, names, seqs)
        db = os.path.join(tmp, "db")
        subprocess.run([KMA, "index", "-i", fa, "-o", db], check=True, stderr=subprocess.DEVNULL)
        cases = {}
        write_fq(os.path.join(OUT, "p33.fq"), 400, 1)
        cases["p33"] = ["-i", "p33.fq"]
        write_fq(os.path.join(OUT, "dos.fq"), 150, 2, eol=b"\r\n")
        cases["dos"] = ["-i", "dos.fq"]
        write_fq(os.path.join(OUT, "p64.fq"), 200, 3, base=64)
        cases["p64"] = ["-i", "p64.fq"]
        write_fa(os.path.join(OUT, "wrap.fa"), 120, 4)
        cases["wrap"] = ["-i", "wrap.fa"]
        write_fq(os.path.join(OUT, "m1.fq"), 250, 5)
        write_fq(os.path.join(OUT, "m2.fq"), 250, 6)
        cases["pe"] = ["-ipe", "m1.fq", "m2.fq"]
        # interleaved input (-int): couples of consecutive records, and a file with an odd number of records (its last one is filed singly)
        write_fq(os.path.join(OUT, "ilv.fq"), 300, 7)
        cases["int"] = ["-int", "ilv.fq"]
        write_fq(os.path.join(OUT, "ilvodd.fq"), 151, 8)
        cases["intodd"] = ["-int", "ilvodd.fq"]
        with open(os.path.join(OUT, "p33.fq"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "p33gz.fq.gz"), "wb", mtime=0) as g:
            shutil.copyfileobj(f, g)
        cases["p33gz"] = ["-i", "p33gz.fq.gz"]
        for case, inp in cases.items():
            for name, flags in SETTINGS.items():
                cmd = [KMA] + inp + ["-o", os.path.join(tmp, "o"), "-t_db", db, "-1t1", "-t", "1", "-s1"] + flags
                r = subprocess.run(cmd, cwd=OUT, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                with gzip.GzipFile(os.path.join(OUT, f"{case}.{name}.s1.gz"), "wb", mtime=0) as g:
                    g.write(r.stdout)
                recs = formats.parse_s1(r.stdout)
                print(case, name, len(recs), "records", [ln for ln in r.stderr.decode().splitlines() if "Phred" in ln])


if __name__ == "__main__":
    main()
