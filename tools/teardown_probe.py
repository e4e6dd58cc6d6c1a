#!/usr/bin/env python3
"""The file-to-file run of 10 M reads (examples/kmahip_map -1t1) three times on the GPU box, the second time with KMAHIP_MAP_TEARDOWN=1:
the process wall beside the program's own stamps, and what its teardown gives back piece by piece.
    gpurun -- 'python3 tools/teardown_probe.py'
"""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from kma_amd import formats, synth
tmp = tempfile.mkdtemp(prefix="td_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k"); formats.write_index(prefix, names, seqs)
fq = os.path.join(tmp, "r.fq")
with open(fq, "wb") as f:
    for a in range(10):
        codes, _, _, _ = synth.make_reads(seqs, 1_000_000, seed=1000 + a)
        part = os.path.join(tmp, "p.fq"); bench.write_fastq_fixed(part, codes); f.write(open(part, "rb").read())
MAP = os.path.join(ROOT, "examples", "kmahip_map")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
for env in ({}, {"KMAHIP_MAP_TEARDOWN": "1"}, {}):
    t0 = time.perf_counter()
    r = subprocess.run([MAP, "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "o"), "-1t1"], env=dict(os.environ, **env), stderr=subprocess.PIPE)
    print(round(time.perf_counter() - t0, 3), env, [l for l in r.stderr.decode().splitlines() if "teardown" in l or "wall" in l][-2:], flush=True)
