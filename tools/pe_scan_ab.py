#!/usr/bin/env python3
"""Stage 2 on the same reads as single reads (scan_se_kernel<., 0, .>: best templates per strand) and as pairs (<., 1, .>: every
candidate, its score and hit count), kernel times by HIP events.  usage (GPU box): python3 tools/pe_scan_ab.py [pairs]"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kma_amd import binding, formats, synth, synth_dev  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp()
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
db = binding.KmaHipDB(prefix)
rd = synth_dev.make_packed_pairs(seqs, n_pairs, seed=7, device=dev)
n = 2 * n_pairs
i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
mate, rc, rc_flag, flag = (i32(n) for _ in range(4))
R_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
T = i32(8 * n)
for mode in ("se", "pe", "se", "pe"):
    db.set_timing(True)
    for k in range(4):
        db.get_timing(k)
    for _ in range(3):
        if mode == "se":
            db.scan_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], rc_flag, flag, R_off, T)
        else:
            db.scan_pe_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], mate, rc, rc_flag, flag, R_off, T)
    torch.cuda.synchronize()
    ms, cnt = db.get_timing(0)
    pms, pcnt = db.get_timing(2)
    db.set_timing(False)
    print(f"{mode}: scan_se_kernel {ms / max(1, cnt):.3f} ms, prefilter {pms / max(1, pcnt):.3f} ms per {n} reads; list entries {int(R_off[-1].item())}", flush=True)
