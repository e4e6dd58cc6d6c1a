#!/usr/bin/env python3
"""The default mode (no -1t1) file to file on n reads: examples/kmahip_map through the batched session and through the one-batch call
(KMAHIP_MAP_ONE_BATCH=1), wall clock seen from outside and peak resident set.   usage (GPU box): python3 tools/e2e_chain.py [reads]"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tmp = tempfile.mkdtemp(prefix="e2ec_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
formats.write_index(prefix, names, seqs)
fq = os.path.join(tmp, "reads.fq")
with open(fq, "wb") as f:
    for a in range(0, n, 2_000_000):
        codes, _, _, _ = synth.make_reads(seqs, min(2_000_000, n - a), seed=1000 + a)
        bench.write_fastq_fixed(os.path.join(tmp, "part.fq"), codes)
        f.write(open(os.path.join(tmp, "part.fq"), "rb").read())
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
mapper = os.path.join(ROOT, "examples", "kmahip_map")
outs = {}
for label, env in (("session", {}), ("one batch", {"KMAHIP_MAP_ONE_BATCH": "1"})):
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([mapper, "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, label.replace(" ", "_"))], stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        dt = time.perf_counter() - t0
        err = r.stderr.decode()
        rss = re.search(r"peak RSS (\d+) MB", err)
        line = [x for x in err.splitlines() if x.startswith("# kmahip_map")]
        print(f"{label}: {dt:.3f} s = {n / dt / 1e6:.2f} M reads/s (rc {r.returncode}), peak RSS {rss.group(1) if rss else '?'} MB | {line[0][:330] if line else ''}", flush=True)
a, b = (os.path.join(tmp, x) for x in ("session", "one_batch"))
same = all(open(a + e, "rb").read() == open(b + e, "rb").read() for e in (".res", ".fsa"))
import gzip  # noqa: E402
same = same and gzip.open(a + ".frag.gz").read() == gzip.open(b + ".frag.gz").read()
print("files identical:", same)
