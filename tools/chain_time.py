#!/usr/bin/env python3
"""The default mode (no -1t1) on the benchmark's shape: stage 2 (kmahip_scan_chain) and the whole run (examples/kmahip_map -chain) on n
150-base reads against the 5 k-gene database, the reference without -1t1 on a sample beside it.
usage (GPU box): python3 tools/chain_time.py [reads [reference sample]]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
tmp = tempfile.mkdtemp(prefix="chain_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
formats.write_index(prefix, names, seqs)
codes, _, _, _ = synth.make_reads(seqs, n, seed=1000)
fq = os.path.join(tmp, "reads.fq")
bench.write_fastq_fixed(fq, codes)
batch = formats.pack_fixed(codes)
db = binding.KmaHipDB(prefix)
db.scan_chain(formats.pack_fixed(codes[:1000]))
for rep in range(2):
    t0 = time.perf_counter()
    o = db.scan_chain(batch)
    dt = time.perf_counter() - t0
    print(f"kmahip_scan_chain (host buffers in and out): {n} reads, {len(o['read'])} records in {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M reads/s", flush=True)
db.close()
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
t0 = time.perf_counter()
r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-chain"], stderr=subprocess.PIPE)
dt = time.perf_counter() - t0
for line in r.stderr.decode().splitlines():
    if "run_chain" in line or "scan_chain" in line:
        print("   ", line)
print(f"kmahip_map -chain, file to file: {dt:.2f} s = {n / dt / 1e6:.2f} M reads/s | {r.stderr.decode().strip().splitlines()[-1] if r.stderr else ''}", flush=True)
kma = os.path.join(ROOT, "oracle", "_ref", "kma")
if os.path.exists(kma) and sample:
    sfq = os.path.join(tmp, "sample.fq")
    rec = os.path.getsize(fq) // n
    with open(fq, "rb") as f, open(sfq, "wb") as g:
        g.write(f.read(rec * min(sample, n)))
    t0 = time.perf_counter()
    subprocess.run([kma, "-i", sfq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    print(f"reference without -1t1, -t 1, {min(sample, n)} reads: {dt:.2f} s = {min(sample, n) / dt / 1e3:.1f} k reads/s", flush=True)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", sfq, "-t_db", prefix, "-o", os.path.join(tmp, "got_s"), "-chain"], check=True, stderr=subprocess.DEVNULL)
    print("  .res of the sample identical:", open(os.path.join(tmp, "got_s.res"), "rb").read() == open(os.path.join(tmp, "ref.res"), "rb").read(), flush=True)
