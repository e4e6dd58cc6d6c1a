#!/usr/bin/env python3
"""`-Mt1 1 -bcNano` file to file on n ONT-like reads of 10 kb against one 5 Mb genome: examples/kmahip_map's wall clock and stamps.
usage (GPU box): python3 tools/e2e_mt1.py [reads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
tmp = tempfile.mkdtemp(prefix="e2em_")
rng = np.random.default_rng(4)
genome = rng.integers(0, 4, 5_000_000, dtype=np.uint8)
prefix = os.path.join(tmp, "g")
formats.write_index(prefix, ["genome5Mb"], [genome])
reads = synth.make_long_reads(genome, n, read_len=10000, seed=8)
fq = os.path.join(tmp, "ont.fq")
synth.write_fastq(fq, reads, prefix="r", qual=b"5")
print(f"{os.path.getsize(fq) / 1e9:.2f} GB of FASTQ", flush=True)
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
mapper = os.path.join(ROOT, "examples", "kmahip_map")
for rep in range(int(os.environ.get("E2E_REPS", "3"))):
    time.sleep(3)          # (the process of the run before is still being taken down behind its parent: GBs of HBM and host pages)
    t0 = time.perf_counter()
    r = subprocess.run([mapper, "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-Mt1", "1", "-bcNano"], stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print(f"{dt:.3f} s = {n / dt / 1e3:.1f} k reads/s (rc {r.returncode})", flush=True)
    for line in r.stderr.decode().splitlines():
        if line.startswith("# kmahip_map") or "ingest:" in line:
            print("   ", line[:520])
