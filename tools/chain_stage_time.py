#!/usr/bin/env python3
"""Stage 2 of the default mode only (kmahip_scan_chain on host buffers), with the kernel stamps of KMAHIP_DEBUG_TIMING; KMAHIP_CHAIN_STOP=2
stops the chaining kernels after the per-strand chaining (no extraction), KMAHIP_CHAIN=slow forces the lane-per-read kernel.
usage (GPU box): [KMAHIP_LIB=...] python3 tools/chain_stage_time.py [reads]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("KMAHIP_CHAIN_TIMING"):
    os.environ["KMAHIP_DEBUG_TIMING"] = "1"          # (KMAHIP_CHAIN_TIMING=1: the fast route as a whole, chunks overlapped, no stamps per kernel)
from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
tmp = tempfile.mkdtemp(prefix="chain_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
formats.write_index(prefix, names, seqs)
codes, _, _, _ = synth.make_reads(seqs, n, seed=1000)
batch = formats.pack_fixed(codes)
db = binding.KmaHipDB(prefix)
db.scan_chain(formats.pack_fixed(codes[:1000]))
for stop in (os.environ.get("CHAIN_STOPS", "0,2").split(",")):
    os.environ["KMAHIP_CHAIN_STOP"] = stop
    print(f"== KMAHIP_CHAIN_STOP={stop} lib={os.environ.get('KMAHIP_LIB', 'default')}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    try:
        o = db.scan_chain(batch)
        print(f"   {len(o['read'])} records, call {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
    except Exception as e:  # noqa: BLE001
        print("   failed:", e, file=sys.stderr, flush=True)
db.close()
