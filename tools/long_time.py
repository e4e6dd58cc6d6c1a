#!/usr/bin/env python3
"""Long noisy reads (the C4 shape, without -Mt1 / -bcNano): N reads of L bases (4 % substitutions, 3 % deletions, 3 % insertions)
against one random genome, stages 2 + 3a through the host-buffer call with kernel timing.
usage (GPU box): python3 tools/long_time.py [reads [read_len [genome_len]]]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
rng = np.random.default_rng(4)
genome = rng.integers(0, 4, G, dtype=np.uint8)
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "g")
t0 = time.perf_counter()
formats.write_index(prefix, ["genome"], [genome])
print(f"index of {G} bp: {time.perf_counter() - t0:.1f} s", flush=True)
reads = synth.make_long_reads(genome, n, read_len=L, seed=8)
if isinstance(reads, tuple):
    reads = reads[0]
b = formats.pack_ragged(reads)
db = binding.KmaHipDB(prefix)
db.set_timing(True)
for label in ("map_se", "map_se again (warm)"):
    db.get_timing(0); db.get_timing(1); db.get_timing(2); db.get_timing(3)
    t0 = time.perf_counter()
    (rc_flag, flag, T_off, T), h = db.map_se(b)
    dt = time.perf_counter() - t0
    ks = [round(db.get_timing(i)[0], 1) for i in (2, 0, 3, 1)]
    bases = int(b.length.sum())
    print(f"{label:22s} {dt * 1e3:9.1f} ms  {n / dt / 1e3:8.1f} k reads/s  {bases / dt / 1e9:6.2f} Gbases/s   kernels prefilter/scan/seed/align ms {ks}", flush=True)
print("mapped", int((h["n_hits"] > 0).sum()), "of", n)
