#!/usr/bin/env python3
"""Wall-clock of the stages beyond the benchmarked step on synthetic 5k-gene data (host-buffer calls, PCIe inclusive):
ConClave, .res statistics, traceback aligner, pile-up + consensus.  usage (GPU box): python3 tools/pipeline_time.py [reads]"""
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
names, seqs = synth.make_gene_db(fam, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
reads, _, _, _ = synth.make_reads(seqs, n, seed=7)
b = formats.pack_fixed(reads)
db = binding.KmaHipDB(prefix)


def timed(label, f):
    t0 = time.perf_counter()
    out = f()
    dt = time.perf_counter() - t0
    print(f"{label:28s} {dt * 1e3:9.1f} ms  ({n / dt / 1e6:7.2f} M reads/s)", flush=True)
    return out


(rc_flag, flag, T_off, T), h = timed("map_se (stages 2 + 3a)", lambda: db.map_se(b))
(rc_flag, flag, T_off, T), h = timed("map_se again (warm)", lambda: db.map_se(b))
cc = timed("conclave_se", lambda: db.conclave_se(b.length, T_off, h))
rows = timed("res_rows", lambda: db.res_rows(cc["w_scores"]))
ok = np.zeros(int(db.info.DB_size), np.uint8)
for r in rows:
    ok[r.template_id] = r.significant
traces = timed("align_trace", lambda: db.align_trace(b, h["rc"], cc["tmpl"], ok))
traces = timed("align_trace again (warm)", lambda: db.align_trace(b, h["rc"], cc["tmpl"], ok))
asm = timed("assemble (pile-up+consensus)", lambda: db.assemble(b, h["rc"], cc["tmpl"], traces))
asm = timed("assemble again (warm)", lambda: db.assemble(b, h["rc"], cc["tmpl"], traces))
print("kept reads", int((traces[0][:, 3] > 0).sum()), "templates assembled", int((asm["asm_len"] > 0).sum()))
