#!/usr/bin/env python3
"""Paired-end path (`-ipe ... -apm p`, BASELINE config C3 shape) through the host-buffer calls, PCIe inclusive:
stage 2 + 3a (map_pe), ConClave over the record slots.  usage (GPU box): python3 tools/pe_time.py [pairs [families [variants]]]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tmp = tempfile.mkdtemp()
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
var = int(sys.argv[3]) if len(sys.argv) > 3 else 5
names, seqs = synth.make_gene_db(fam, var, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
m1, m2, _ = synth.make_pairs(seqs, n, seed=11)
inter = np.empty((2 * n, m1.shape[1]), np.uint8)
inter[0::2] = m1
inter[1::2] = m2
b = formats.pack_fixed(inter)
db = binding.KmaHipDB(prefix)
for label in ("map_pe (stages 2 + 3a)", "map_pe again (warm)"):
    t0 = time.perf_counter()
    (mate, rc, rc_flag, flag, R_off, T), h = db.map_pe(b)
    dt = time.perf_counter() - t0
    print(f"{label:28s} {dt * 1e3:9.1f} ms  ({n / dt / 1e6:7.2f} M pairs/s)", flush=True)
db.set_timing(True)
db.get_timing(0)
(mate, rc, rc_flag, flag, R_off, T), h = db.map_pe(b)
ks = {name: db.get_timing(i) for i, name in ((2, "scan_prefilter_kernel"), (0, "scan_se_kernel (all-candidates mode)"), (3, "seed_tasks_kernel"), (1, "align_tasks_kernel"))}
print("kernel times of one map_pe call:", ", ".join(f"{k} {v[0]:.2f} ms" for k, v in ks.items()), flush=True)
db.set_timing(False)
t0 = time.perf_counter()
cc = db.conclave_pe(b.length, mate, R_off, h)
dt = time.perf_counter() - t0
print(f"{'conclave_pe':28s} {dt * 1e3:9.1f} ms")
print(f"candidate (record, template) tasks: {int(R_off[-1])} for {n} pairs; summed hits kept: {int(h['n_hits'].sum())}")
kinds = np.bincount(h["kind"], minlength=5)
print("pair kinds (0 singly / none, 1 proper, 2 unmated, 3 first only, 4 second only):", kinds.tolist())
print("templates with score:", int((cc["w_scores"] > 0).sum()))

# DP problem histogram of the paired path (diagnostic build only: KMAHIP_LIB=kma_amd/libkmahip_diag.so)
import ctypes
L = binding.lib()
if hasattr(L, "kmahip_diag_hist"):
    buf = (ctypes.c_uint64 * 256)()
    L.kmahip_diag_hist(buf, 1)
    db.map_pe(b)                 # production kernels (seeded by seed_tasks_kernel): only the queue counters tick
    L.kmahip_diag_hist(buf, 0)
    v = list(buf)
    print("production launch: queued tiny / narrow / wide:", v[200:203], " queue full:", v[204:207], " solved outside the queues:", v[208],
          "cells there:", v[209], "of them >= 64 columns:", v[210])
    L.kmahip_diag_hist(buf, 1)
    db.set_stats(True)
    db.map_pe(b)
    db.set_stats(False)
    L.kmahip_diag_hist(buf, 0)
    v = list(buf)
    print("DP calls by q_len :", {i: v[i] for i in range(64) if v[i]})
    print("DP calls by mode k:", {i - 2: v[128 + i] for i in range(5)})
    print("queued tiny / narrow / wide:", v[200:203], " queue full:", v[204:207], " solved outside the queues:", v[208])
    print("DP calls by t_len/4:", {4 * i: v[136 + i] for i in range(64) if v[136 + i]})
