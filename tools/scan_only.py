#!/usr/bin/env python3
"""Runs only the stage-2 scan (and optionally stage 3a) a few times on 2M synthetic reads: target for profilers.
usage (GPU box): [KMAHIP_LIB=...] python3 tools/scan_only.py [reads] [reps] [scan|align|both]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kma_amd import binding, formats, synth, synth_dev  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
what = sys.argv[3] if len(sys.argv) > 3 else "scan"
tmp = tempfile.mkdtemp()
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
db = binding.KmaHipDB(prefix)
dev = torch.device("cuda", 0)
rd = synth_dev.make_packed_reads(seqs, n, seed=1000, device=dev)
i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
rc_flag, flag, T_off, T = i32(n), i32(n), torch.empty(n + 1, dtype=torch.int64, device=dev), i32(8 * n)
n_hits, best, oflag = i32(n), i32(n), i32(n)
h = [i32(8 * n) for _ in range(4)]
aln = torch.zeros(int(db.info.DB_size), dtype=torch.int64, device=dev)
uniq = torch.zeros_like(aln)
db.set_timing(True)
for _ in range(reps):
    if what in ("scan", "both") or _ == 0:
        db.scan_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], rc_flag, flag, T_off, T)
    if what in ("align", "both"):
        db.align_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], 150, rc_flag, flag, T_off, T,
                        n_hits, best, oflag, *h, aln, uniq)
torch.cuda.synchronize()
s_ms, s_n = db.get_timing(0)
a_ms, a_n = db.get_timing(1)
p_ms, p_n = db.get_timing(2)
k_ms, k_n = db.get_timing(3)
print(f"reads {n}: prefilter {p_ms / max(1, p_n):.3f} ms x{p_n}, scan {s_ms / max(1, s_n):.3f} ms x{s_n}, "
      f"seed {k_ms / max(1, k_n):.3f} ms x{k_n}, align {a_ms / max(1, a_n):.3f} ms x{a_n}")
