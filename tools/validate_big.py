#!/usr/bin/env python3
"""One-off large validation on the GPU box: > maxFrag (1 M) reads incl. reads with indels against a 500-gene database, the
compiled reference (oracle/_ref/kma -1t1 -t 1) and the library side by side; the two `.res` files and consensus FASTAs must be
identical. Exercises the chunked read order of the pile-up (insertion columns) at scale.
usage: python3 tools/validate_big.py [reads [families [variants [max_div]]]]   (C5-like: 2000000 5000 10 0.045)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import golden_util  # noqa: E402
from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp()
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 100
var = int(sys.argv[3]) if len(sys.argv) > 3 else 5
div = float(sys.argv[4]) if len(sys.argv) > 4 else 0.04
names, seqs = synth.make_gene_db(fam, var, 600, 1500, div, seed=777)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
base, _, _, _ = synth.make_reads(seqs, n, seed=99)
rng = np.random.default_rng(5)
reads = [r for r in base]
indel_div = int(os.environ.get("KMAHIP_VALIDATE_INDEL_DIV", "40"))     # every 40th read (2.5 %) gets an indel or two
for i in rng.choice(n, size=(n // indel_div if indel_div > 0 else 0), replace=False):
    r = reads[i]
    parts, j = [], 0
    while j < len(r):
        e = min(len(r), j + int(rng.integers(30, 120)))
        parts.append(r[j:e])
        u = rng.random()
        if u < 0.45:
            parts.append(rng.integers(0, 4, int(rng.integers(1, 4)), dtype=np.uint8))
        elif u < 0.9:
            e = min(len(r), e + int(rng.integers(1, 4)))
        j = e
    reads[i] = np.concatenate(parts)
fq = os.path.join(tmp, "reads.fq")
synth.write_fastq(fq, reads, lens=None)
t0 = time.perf_counter()
subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-1t1", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
t_ref = time.perf_counter() - t0
b = formats.pack_ragged(reads)
db = binding.KmaHipDB(prefix)
t0 = time.perf_counter()
(rc_flag, flag, T_off, T), h = db.map_se(b)
cc = db.conclave_se(b.length, T_off, h)
rows = db.res_rows(cc["w_scores"])
ok = np.zeros(int(db.info.DB_size), np.uint8)
for r in rows:
    ok[r.template_id] = r.significant
traces = db.align_trace(b, h["rc"], cc["tmpl"], ok)
asm = db.assemble(b, h["rc"], cc["tmpl"], traces, consensus=True)
t_ours = time.perf_counter() - t0
lines = ["#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n"]
fsa = []
for r in rows:
    if r.significant:
        t = r.template_id
        line = db.res_line(names[t - 1], r, asm["cover"][t], asm["aln_len"][t], asm["depth"][t])
        if line:
            lines.append(line)
            fsa.append((names[t - 1], asm["consensus"][t]))
ref_res = open(os.path.join(tmp, "ref.res")).read()
ref_fsa = open(os.path.join(tmp, "ref.fsa")).read()
ins_reads = int(sum(1 for i in range(b.n) if traces[0][i, 7] > 0))
print(f"db {fam} families x {var} variants; reads {n}, reads aligned with an insertion {ins_reads}, .res rows {len(lines) - 1}")
print(f"reference {t_ref:.1f} s (whole pipeline, 1 thread), library {t_ours:.2f} s (host-buffer calls incl. staging)")
print("res identical:", "".join(lines) == ref_res, " consensus FASTA identical:", golden_util.fsa_text(fsa) == ref_fsa)
# the same through the one-call pipeline (reads uploaded once)
t0 = time.perf_counter()
o = db.run_se(b, per_read=False)
t_one = time.perf_counter() - t0
lines1, fsa1 = [lines[0]], []
for r in o["rows"]:
    if r.significant:
        t = r.template_id
        line = db.res_line(names[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
        if line:
            lines1.append(line)
            fsa1.append((names[t - 1], o["consensus"][t]))
print(f"kmahip_run_se: {t_one:.3f} s  (upload {o['ms'][0]:.0f} ms, stages 2+3a {o['ms'][1]:.0f}, ConClave+stats {o['ms'][2]:.0f}, traceback {o['ms'][3]:.0f}, "
      f"pile-up+consensus {o['ms'][4]:.0f}, copies {o['ms'][5]:.0f});  res identical: {''.join(lines1) == ref_res}  consensus identical: {golden_util.fsa_text(fsa1) == ref_fsa}")
if "".join(lines1) != ref_res:
    sys.exit(1)
if "".join(lines) != ref_res:
    for a, c in zip("".join(lines).splitlines(), ref_res.splitlines()):
        if a != c:
            print("ours:", a)
            print("ref :", c)
            break
    sys.exit(1)

# diagnostic build only (KMAHIP_LIB=kma_amd/libkmahip_diag.so): where the DP problems of stage 3a went
import ctypes
L = binding.lib()
if hasattr(L, "kmahip_diag_hist"):
    buf = (ctypes.c_uint64 * 256)()
    L.kmahip_diag_hist(buf, 1)
    db.set_timing(True)
    db.map_se(b)
    v = list(buf) if not L.kmahip_diag_hist(buf, 0) else list(buf)
    print("stage 3a DP problems: queued tiny / narrow / wide:", v[200:203], " queue full:", v[204:207], " solved outside the queues:", v[208],
          "cells there:", v[209], "of them >= 64 columns:", v[210])
    print("kernel ms (scan, align, prefilter, seed):", [round(db.get_timing(i)[0], 2) for i in range(4)])
    L.kmahip_diag_hist(buf, 1)
    db.set_stats(True)
    db.map_se(b)
    db.set_stats(False)
    L.kmahip_diag_hist(buf, 0)
    v = list(buf)
    print("DP calls by q_len :", {i: v[i] for i in range(64) if v[i]})
    print("DP cells by q_len :", {i: v[64 + i] for i in range(64) if v[64 + i]})
    print("DP calls by t_len/4:", {4 * i: v[136 + i] for i in range(64) if v[136 + i]})
    t_id = int(os.environ.get("KMAHIP_TASK", "-1"))
    if t_id >= 0:
        (rc_flag2, flag2, T_off2, T2), h2 = db.map_se(b)
        r_id = int(np.searchsorted(T_off2, t_id, side="right") - 1)
        print(f"task {t_id}: read {r_id}, length {int(b.length[r_id])}, rc_flag {int(rc_flag2[r_id])}, flag {int(flag2[r_id])}, candidates "
              f"{T2[T_off2[r_id]:T_off2[r_id + 1]].tolist()}, this one {int(T2[t_id])}, template length {int(formats.read_lengths(prefix)[abs(int(T2[t_id]))])}, "
              f"hits {int(h2['n_hits'][r_id])}, best {int(h2['best_score'][r_id])}")
        print("read:", "".join("ACGTN"[c] for c in reads[r_id]))
        w0 = t_id & ~63
        for tt in range(w0, w0 + 64):
            rr = int(np.searchsorted(T_off2, tt, side="right") - 1)
            if int(rc_flag2[rr]) < 140 or int(b.length[rr]) != 150:
                print(f"  task {tt}: read {rr} len {int(b.length[rr])} rc_flag {int(rc_flag2[rr])} tmpl {int(T2[tt])} hits {int(h2['n_hits'][rr])} best {int(h2['best_score'][rr])}",
                      "".join("ACGTN"[c] for c in reads[rr]))
        # which read of that wavefront is the slow one: each alone, align kernel time
        slow = []
        for rr in sorted({int(np.searchsorted(T_off2, tt, side="right") - 1) for tt in range(w0, w0 + 64)}):
            one = formats.pack_ragged([reads[rr]])
            db.get_timing(1); db.get_timing(3)
            db.map_se(one)
            slow.append((round(db.get_timing(1)[0], 3), rr, len(reads[rr])))
        slow.sort(reverse=True)
        print("slowest single-read align kernels (ms, read, length):", slow[:6])
        one = formats.pack_ragged([reads[slow[0][1]]])
        L.kmahip_diag_hist(buf, 1)
        db.set_stats(True)
        db.map_se(one)
        db.set_stats(False)
        L.kmahip_diag_hist(buf, 0)
        v = list(buf)
        print("slowest read alone: DP calls by q_len", {i: v[i] for i in range(64) if v[i]}, "cells", {i: v[64 + i] for i in range(64) if v[64 + i]},
              "by t_len/4", {4 * i: v[136 + i] for i in range(64) if v[136 + i]}, "mode", {i - 2: v[128 + i] for i in range(5) if v[128 + i]},
              "queued", v[200:203], "full", v[204:207], "outside", v[208], v[209], v[210], "align stats", [getattr(db.get_align_stats(), f) for f in ("lookups", "mem_bases", "dp_cells", "tasks")])
