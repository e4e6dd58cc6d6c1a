#!/usr/bin/env python3
"""File to file on the GPU box: FASTQ -> .res / .fsa / .frag.gz through examples/kmahip_map (whole-process wall clock, incl. HIP
start-up, kmahip_db_open, ingest, the device run and the three writers), and the reference binary on a subset for parity and
its rates (-t 1, -t nproc, nproc independent -t 1 processes over read shards).
usage: python3 tools/e2e_time.py [reads [sample [gz]]]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
gz = len(sys.argv) > 3 and sys.argv[3] == "gz"
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
MAP = os.path.join(ROOT, "examples", "kmahip_map")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
tmp = tempfile.mkdtemp(prefix="e2e_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
t0 = time.perf_counter()
if os.path.exists(KMA):
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
else:
    formats.write_index(prefix, names, seqs)
print(f"index: {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
fq = os.path.join(tmp, "reads.fq")
with open(fq, "wb") as f:
    for a in range(0, n, 2_000_000):
        m = min(2_000_000, n - a)
        codes, _, _, _ = synth.make_reads(seqs, m, seed=1000 + a)
        sub = os.path.join(tmp, "part.fq")
        bench.write_fastq_fixed(sub, codes)
        # (names restart per part; the reference does not care, and both sides see the same file)
        f.write(open(sub, "rb").read())
        os.unlink(sub)
print(f"FASTQ of {n} reads ({os.path.getsize(fq) / 1e9:.2f} GB): {time.perf_counter() - t0:.1f} s", flush=True)
inp = fq
if gz:
    t0 = time.perf_counter()
    subprocess.run(["gzip", "-1", "-k", fq], check=True)
    inp = fq + ".gz"
    print(f"gzip -1: {time.perf_counter() - t0:.1f} s ({os.path.getsize(inp) / 1e9:.2f} GB)", flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([MAP, "-i", inp, "-t_db", prefix, "-o", os.path.join(tmp, "got")], stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print(f"kmahip_map e2e run {rep}: {dt:.2f} s = {n / dt / 1e6:.2f} M reads/s   rc {r.returncode}  | {r.stderr.decode().strip().splitlines()[-1] if r.stderr else ''}", flush=True)
if os.path.exists(KMA) and sample:
    m = min(sample, n)
    sfq = os.path.join(tmp, "sample.fq")
    rec = os.path.getsize(fq) // n
    with open(fq, "rb") as f, open(sfq, "wb") as g:
        g.write(f.read(rec * m))
    nproc = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    subprocess.run([KMA, "-i", sfq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-1t1", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    subprocess.run([KMA, "-i", sfq, "-o", os.path.join(tmp, "reft"), "-t_db", prefix, "-1t1", "-t", str(nproc)], check=True, stderr=subprocess.DEVNULL)
    tn = time.perf_counter() - t0
    # nproc independent processes over equal shards (wall = slowest)
    per = (m + nproc - 1) // nproc
    shards = []
    with open(sfq, "rb") as f:
        for i in range(nproc):
            p = os.path.join(tmp, f"shard{i}.fq")
            with open(p, "wb") as g:
                g.write(f.read(rec * per))
            shards.append(p)
    t0 = time.perf_counter()
    procs = [subprocess.Popen([KMA, "-i", p, "-o", p + ".out", "-t_db", prefix, "-1t1", "-t", "1"], stderr=subprocess.DEVNULL) for p in shards]
    for p in procs:
        p.wait()
    ts = time.perf_counter() - t0
    subprocess.run([MAP, "-i", sfq, "-t_db", prefix, "-o", os.path.join(tmp, "gots")], check=True, stderr=subprocess.DEVNULL)
    same = open(os.path.join(tmp, "gots.res"), "rb").read() == open(os.path.join(tmp, "ref.res"), "rb").read()
    cpu = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?"
    print(f"reference on {m} reads ({cpu}, {nproc} cores available): -t 1 {t1:.1f} s = {m / t1 / 1e3:.1f} k reads/s; -t {nproc} {tn:.1f} s = {m / tn / 1e3:.1f} k reads/s; "
          f"{nproc} processes x -t 1 over shards {ts:.1f} s = {m / ts / 1e3:.1f} k reads/s; .res of the sample identical: {same}", flush=True)
