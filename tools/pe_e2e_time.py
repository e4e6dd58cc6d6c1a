#!/usr/bin/env python3
"""Paired end (-ipe r1 r2 -apm p -1t1) file to file: examples/kmahip_map on n pairs of the benchmark's shape, the reference on a sample.
usage (GPU box): python3 tools/pe_e2e_time.py [pairs [reference sample]]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
tmp = tempfile.mkdtemp(prefix="pe_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
formats.write_index(prefix, names, seqs)
lut = np.frombuffer(b"ACGT", np.uint8)
paths = [os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")]
with open(paths[0], "wb") as f1, open(paths[1], "wb") as f2:
    for a in range(0, n, 500_000):
        m = min(500_000, n - a)
        m1, m2, _ = synth.make_pairs(seqs, m, seed=77 + a)
        for f, ms, tag in ((f1, m1, b"/1"), (f2, m2, b"/2")):
            L = ms.shape[1]
            rec = np.empty((m, 1 + 10 + 2 + 1 + L + 3 + L + 1), np.uint8)
            rec[:, 0] = ord("@"); rec[:, 1] = ord("p")
            idx = np.arange(a, a + m)
            for d in range(9):
                rec[:, 10 - d] = ord("0") + (idx // 10 ** d) % 10
            rec[:, 11:13] = np.frombuffer(tag, np.uint8)
            o = 13
            rec[:, o] = 10; o += 1
            rec[:, o:o + L] = lut[ms]; o += L
            rec[:, o:o + 3] = np.frombuffer(b"\n+\n", np.uint8); o += 3
            rec[:, o:o + L] = ord("I"); o += L
            rec[:, o] = 10
            f.write(rec.tobytes())
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", paths[0], paths[1], "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-1t1", "-apm", "p"], stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print(f"kmahip_map -ipe, file to file: {n} pairs in {dt:.2f} s = {2 * n / dt / 1e6:.2f} M reads/s | {r.stderr.decode().strip().splitlines()[-1] if r.stderr else ''}", flush=True)
    if rep and os.environ.get("KMAHIP_DEBUG_TIMING"):
        print(r.stderr.decode(), flush=True)
kma = os.path.join(ROOT, "oracle", "_ref", "kma")
if os.path.exists(kma) and sample:
    sub = []
    for p in paths:
        rec = os.path.getsize(p) // n
        s = p + ".sample"
        with open(p, "rb") as f, open(s, "wb") as g:
            g.write(f.read(rec * min(sample, n)))
        sub.append(s)
    t0 = time.perf_counter()
    subprocess.run([kma, "-ipe", sub[0], sub[1], "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-1t1", "-apm", "p", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    print(f"reference -ipe -apm p -1t1 -t 1, {min(sample, n)} pairs: {dt:.2f} s = {2 * min(sample, n) / dt / 1e3:.1f} k reads/s", flush=True)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", sub[0], sub[1], "-t_db", prefix, "-o", os.path.join(tmp, "got_s"), "-1t1", "-apm", "p"], check=True, stderr=subprocess.DEVNULL)
    print("  .res of the sample identical:", open(os.path.join(tmp, "got_s.res"), "rb").read() == open(os.path.join(tmp, "ref.res"), "rb").read(), flush=True)
