#!/usr/bin/env python3
"""Paired end file to file (-ipe r1 r2 -apm p -1t1) on n pairs of 2 x 150 nt: examples/kmahip_map's wall clock and its own stamps.
usage (GPU box): python3 tools/e2e_pe.py [pairs]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
tmp = tempfile.mkdtemp(prefix="e2ep_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
formats.write_index(prefix, names, seqs)
r1, r2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
with open(r1, "wb") as f1, open(r2, "wb") as f2:
    for a in range(0, n, 1_000_000):
        m = min(1_000_000, n - a)
        c1, c2 = synth.make_pairs(seqs, m, seed=2000 + a)[:2]
        for codes, f in ((c1, f1), (c2, f2)):
            bench.write_fastq_fixed(os.path.join(tmp, "part.fq"), codes)
            f.write(open(os.path.join(tmp, "part.fq"), "rb").read())
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
mapper = os.path.join(ROOT, "examples", "kmahip_map")
for rep in range(3):
    t0 = time.perf_counter()
    r = subprocess.run([mapper, "-ipe", r1, r2, "-apm", "p", "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-1t1"], stderr=subprocess.PIPE,
                       env=dict(os.environ, KMAHIP_DEBUG_TIMING="1" if rep == 2 else ""))
    dt = time.perf_counter() - t0
    lines = r.stderr.decode().splitlines()
    print(f"{dt:.3f} s = {2 * n / dt / 1e6:.2f} M reads/s (rc {r.returncode})", flush=True)
    for line in lines:
        if line.startswith("# kmahip_map") or (rep == 2 and ("run_pe" in line or "frag" in line)):
            print("   ", line[:420])
