#!/usr/bin/env python3
"""Peak resident set of examples/kmahip_map (the batched session) against the size of its input: the same reads 1x, 2x and 4x.
usage (GPU box): python3 tools/rss_check.py [reads of the 1x file]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tmp = tempfile.mkdtemp(prefix="rss_")
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
for mult in (1, 2, 4):
    big = fq
    if mult > 1:
        big = os.path.join(tmp, f"x{mult}.fq")
        with open(big, "wb") as f:
            for _ in range(mult):
                with open(fq, "rb") as g:
                    while True:
                        blk = g.read(1 << 26)
                        if not blk:
                            break
                        f.write(blk)
    for mode, env in (("session", {}), ("one batch", {"KMAHIP_MAP_ONE_BATCH": "1"})):
        if mode == "one batch" and mult > 2:
            continue
        r = subprocess.run([mapper, "-i", big, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-1t1"], stderr=subprocess.PIPE, env=dict(os.environ, **env))
        last = r.stderr.decode().strip().splitlines()[-1] if r.stderr else ""
        m = re.search(r"peak RSS (\d+) MB", last)
        print(f"{mult * n} reads, {mode}: rc {r.returncode}, peak RSS {m.group(1) if m else '?'} MB | {last[:160]}", flush=True)
    if mult > 1:
        os.unlink(big)
