#!/usr/bin/env python3
"""BASELINE config C4 at its own size on the GPU box: bench.py's C4 leg alone (reads generated on the device).
usage: python3 tools/c4_time.py [reads [parity_reads]]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
parity = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with tempfile.TemporaryDirectory() as tmp:
    out = bench.c4_leg(tmp, n, 0, parity)
print(json.dumps(out))
