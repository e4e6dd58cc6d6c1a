#!/usr/bin/env python3
"""Golden taps of KMA's DEFAULT template finder (no -1t1: save_kmers_chain, savekmers.c:5127-5945) on the reads and indexes of
the committed `se` and `long` fixtures, written by the compiled reference (oracle/_ref/kma):
    s2_chain.bin.gz     the S2 stream (`-s2`): one record per accepted chain, query bounds appended to the header
    chain.res, chain.frag.gz, chain.fsa.gz    the final files of the same run
usage: python3 tests/golden/make_golden_chain.py"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util  # noqa: E402

KMA = os.path.join(ROOT, "oracle", "_ref", "kma")

for name in ("se", "long"):
    with tempfile.TemporaryDirectory() as tmp:
        g = golden_util.load_se(tmp, name)
        fq = os.path.join(tmp, "reads.fq")
        with gzip.open(os.path.join(HERE, name, "reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
            shutil.copyfileobj(f, o)
        base = [KMA, "-i", fq, "-o", os.path.join(tmp, "out"), "-t_db", g["prefix"], "-t", "1"]
        with open(os.path.join(tmp, "s2.bin"), "wb") as f:
            subprocess.run(base + ["-s2"], check=True, stdout=f, stderr=subprocess.DEVNULL)
        subprocess.run(base, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dst = os.path.join(HERE, name)
        with open(os.path.join(tmp, "s2.bin"), "rb") as f, gzip.open(os.path.join(dst, "s2_chain.bin.gz"), "wb", 9, mtime=0) if False else gzip.GzipFile(os.path.join(dst, "s2_chain.bin.gz"), "wb", 9, mtime=0) as o:
            shutil.copyfileobj(f, o)
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(dst, "chain.res"))
        shutil.copy(os.path.join(tmp, "out.frag.gz"), os.path.join(dst, "chain.frag.gz"))
        with open(os.path.join(tmp, "out.fsa"), "rb") as f, gzip.GzipFile(os.path.join(dst, "chain.fsa.gz"), "wb", 9, mtime=0) as o:
            shutil.copyfileobj(f, o)
        print(name, os.path.getsize(os.path.join(dst, "s2_chain.bin.gz")), "bytes of S2 tap")
