#!/usr/bin/env python3
"""Timing ablations of the two main kernels (diagnostic build libkmahip_diag.so; results are wrong by design).
usage (GPU box): KMAHIP_LIB=kma_amd/libkmahip_diag.so python tools/ablate.py [reads]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KMAHIP_LIB", os.path.join(ROOT, "kma_amd", "libkmahip_diag.so"))
import torch  # noqa: E402

from kma_amd import binding, formats, synth, synth_dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
tmp = tempfile.mkdtemp()
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
db = binding.KmaHipDB(prefix)
dev = torch.device("cuda", 0)
hard = bool(os.environ.get("KMAHIP_HARD"))          # 2 % unmappable reads + 2 % reads with 60-120 foreign end bases
rd = synth_dev.make_packed_reads(seqs, n, seed=1000, device=dev, random_frac=0.02 if hard else 0.0, junk_frac=0.02 if hard else 0.0)
i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
rc_flag, flag, T_off, T = i32(n), i32(n), torch.empty(n + 1, dtype=torch.int64, device=dev), i32(8 * n)
n_hits, best, oflag = i32(n), i32(n), i32(n)
h = [i32(8 * n) for _ in range(4)]
aln = torch.zeros(int(db.info.DB_size), dtype=torch.int64, device=dev)
uniq = torch.zeros_like(aln)


def run(scan_abl, align_abl, reps=3):
    os.environ["KMAHIP_ABLATE_SCAN"] = str(scan_abl)
    os.environ["KMAHIP_ABLATE_ALIGN"] = str(align_abl)
    db.set_timing(True)
    for _ in range(reps + 1):
        if scan_abl >= 0:
            db.scan_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], rc_flag, flag, T_off, T)
        if align_abl >= 0:
            db.align_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], 150, rc_flag, flag, T_off, T,
                            n_hits, best, oflag, *h, aln, uniq)
    torch.cuda.synchronize()
    s_ms, s_n = db.get_timing(0)
    a_ms, a_n = db.get_timing(1)
    p_ms, p_n = db.get_timing(2)
    k_ms, k_n = db.get_timing(3)
    return (s_ms / max(1, s_n), a_ms / max(1, a_n) + k_ms / max(1, k_n), p_ms / max(1, p_n))


print(f"reads {n}")
for name, sa in (("scan full", 0), ("scan no-machines", 1), ("scan no-probe(phase1)", 2), ("scan no-prefilter-probe", 4),
                 ("scan no-probe no-machines", 3), ("scan nothing", 7), ("scan no-2b", 8), ("scan no-finish", 16),
                 ("scan 2a loads only", 32), ("scan no-2b no-finish", 24), ("scan 2a-loads-only no-2b no-finish", 56)):
    t = run(sa, -1)
    print(f"{name:32s} {t[0] + t[2]:8.2f} ms  (prefilter {t[2]:.2f} + scan {t[0]:.2f})")
run(0, -1)  # restore real candidates
for name, aa in (("align full", 0), ("align no-DP", 1), ("align no-wide(q>16)", 8), ("align no-1x1", 16), ("align no-2..16", 32),
                 ("align only-1x1", 40), ("align only-2..16", 24), ("align only-wide", 48), ("align no-coop", 64), ("align no-chain+", 4), ("align no-seed+", 2), ("align nothing outside the queues", 128)):
    print(f"{name:32s} {run(-1, aa)[1]:8.2f} ms")

# DP problem histogram (stats launch)
import ctypes
L = binding.lib()
if hasattr(L, "kmahip_diag_hist"):
    buf = (ctypes.c_uint64 * 256)()
    L.kmahip_diag_hist(buf, 1)
    os.environ["KMAHIP_ABLATE_SCAN"] = "0"; os.environ["KMAHIP_ABLATE_ALIGN"] = "0"
    db.set_stats(True)
    db.scan_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], rc_flag, flag, T_off, T)
    db.align_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], 150, rc_flag, flag, T_off, T,
                    n_hits, best, oflag, *h, aln, uniq)
    torch.cuda.synchronize()
    db.set_stats(False)
    L.kmahip_diag_hist(buf, 0)
    v = list(buf)
    print("tasks", int(T_off[-1].item()))
    print("calls by q_len :", {i: v[i] for i in range(64) if v[i]})
    print("cells by q_len :", {i: v[64 + i] for i in range(64) if v[64 + i]})
    print("queued tiny / narrow / wide:", v[200:203], " queue full:", v[204:207], " solved outside the queues:", v[208], "cells there:", v[209],
          "of them >= 64 columns:", v[210])
    print("calls by mode k:", {i - 2: v[128 + i] for i in range(5)})
    print("calls by t_len/4:", {4 * i: v[136 + i] for i in range(64) if v[136 + i]})
