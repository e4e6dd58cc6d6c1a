#!/usr/bin/env python3
"""Stage 2 of the default mode on long reads (kmahip_scan_chain: the lane-per-read route), with the stamps of KMAHIP_DEBUG_TIMING and
KMAHIP_CHAIN_STOP = 0 / 1 (anchors only) / 2 (+ chaining) / 3 (+ templates). usage (GPU box): python3 tools/chain_long_time.py [reads [length]]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KMAHIP_DEBUG_TIMING"] = "1"
import numpy as np  # noqa: E402
from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
tmp = tempfile.mkdtemp(prefix="chainl_")
rng = np.random.default_rng(4)
genome = rng.integers(0, 4, 2_000_000, dtype=np.uint8)
prefix = os.path.join(tmp, "g")
formats.write_index(prefix, ["genome2Mb"], [genome])
reads = synth.make_long_reads(genome, n, read_len=L, seed=8)
batch = formats.pack_ragged(reads)
db = binding.KmaHipDB(prefix)
db.scan_chain(formats.pack_ragged(reads[:64]))
for stop in os.environ.get("CHAIN_STOPS", "0,1,2,3").split(","):
    os.environ["KMAHIP_CHAIN_STOP"] = stop
    print(f"== KMAHIP_CHAIN_STOP={stop}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    try:
        o = db.scan_chain(batch)
        print(f"   {len(o['read'])} records, call {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
    except Exception as e:  # noqa: BLE001
        print("   failed:", e, file=sys.stderr, flush=True)
db.close()
