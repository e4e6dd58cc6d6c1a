#!/usr/bin/env python3
"""C4-shaped run (tools/mt1_time.py's workload) repeated in ONE process: the trace stage's time call by call -- how much of the spread
between runs is the process, how much the call.  usage (GPU box): python3 tools/mt1_repeat.py [reads [calls]]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import binding, formats, synth_dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(4)
genome = rng.integers(0, 4, 5_000_000, dtype=np.uint8)
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "g")
formats.write_index(prefix, ["genome"], [genome])
rd = synth_dev.make_long_reads_packed(genome, n, read_len=10000, seed=8, device="cuda:0", keep_codes=0)
b = formats.ReadBatch(rd["seq"], rd["seq_off"], rd["length"], rd["N"][:0], rd["N_off"])
db = binding.KmaHipDB(prefix)
out = []
for i in range(calls):
    t0 = time.perf_counter()
    o = db.run_mt1(b, 1, consensus=False)
    out.append((round(o["ms"][0], 1), round(o["ms"][3], 1), round(1e3 * (time.perf_counter() - t0), 1)))
print("upload ms / trace ms / call ms:", out)
db.close()
