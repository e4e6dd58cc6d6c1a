#!/usr/bin/env python3
"""Where the file-to-file second goes: examples/kmahip_map on n reads with the library's stamps, the wall clock of the child from
fork to exit seen from outside, and the same for a run that exits right after the index is open (KMAHIP_MAP_STOP=open).
usage (GPU box): python3 tools/e2e_breakdown.py [reads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tmp = tempfile.mkdtemp(prefix="e2e_")
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
for rep in range(3):
    for env_extra, label in (({}, "full run"), ({"KMAHIP_MAP_STOP": "open"}, "stop after open"), ({"KMAHIP_MAP_STOP": "run"}, "stop after the device run")):
        t0 = time.perf_counter()
        r = subprocess.run([mapper, "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-1t1"] + sys.argv[2:], stderr=subprocess.PIPE,
                           env=dict(os.environ, KMAHIP_DEBUG_TIMING="1", **env_extra))
        dt = time.perf_counter() - t0
        lines = r.stderr.decode().splitlines()
        print(f"== {label}: {dt:.3f} s seen from outside (rc {r.returncode})")
        if rep == 0:
            for line in lines:
                if any(k in line for k in ("kmahip_map", "write_rows", "frag_write", "db_open", "ingest:", "run_se", "session")):
                    print("   ", line[:400])
        elif lines:
            print("   ", lines[-1][:400])
