#!/usr/bin/env python3
"""File to file on the GPU box, outside bench.py: FASTQ -> .res / .fsa / .frag.gz through examples/kmahip_map (whole-process wall
clock), plain and gzip-compressed, and the reference binary on a sample of the same file (bench.e2e_leg does the work).
usage: python3 tools/e2e_time.py [reads [sample]]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
tmp = tempfile.mkdtemp(prefix="e2e_")
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db5k")
formats.write_index(prefix, names, seqs)
e2e, ref = bench.e2e_leg(tmp, prefix, seqs, n, sample, log=lambda s: print(s, flush=True))
print(json.dumps({"e2e": e2e, "reference": ref}, indent=1))
