#!/usr/bin/env python3
"""BASELINE config C4 on the GPU box: N ONT-like reads of L bases (4 % substitutions, 3 % deletions, 3 % insertions) against ONE random
genome, `-Mt1 1 -bcNano` through kmahip_run_mt1; the reference binary (oracle/_ref/kma, one thread) on the first `check` reads for the
parity of .res / consensus / fragment rows and for its time.
usage: python3 tools/mt1_time.py [reads [read_len [genome_len [check]]]]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from kma_amd import binding, formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
check = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
rng = np.random.default_rng(4)
genome = rng.integers(0, 4, G, dtype=np.uint8)
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "g")
t0 = time.perf_counter()
synth.write_fasta(prefix + ".fsa", ["genome"], [genome])
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print(f"kma index of {G} bp: {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
reads = synth.make_long_reads(genome, n, read_len=L, seed=8)
b = formats.pack_ragged(reads)
print(f"{n} reads generated + packed: {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
db = binding.KmaHipDB(prefix)
print(f"kmahip_db_open: {time.perf_counter() - t0:.1f} s ({db.info.total_bytes / 1e6:.0f} MB in HBM)", flush=True)
bases = int(b.length.sum())
for label in ("run_mt1", "run_mt1 again (warm)"):
    t0 = time.perf_counter()
    o = db.run_mt1(b, 1, consensus=False)
    dt = time.perf_counter() - t0
    print(f"{label:22s} {dt * 1e3:9.1f} ms  {n / dt:9.0f} reads/s  {bases / dt / 1e9:6.3f} Gbases/s   ms upload/-/stats/trace/pile-up+consensus/copies {[round(x, 1) for x in o['ms']]}", flush=True)
print("kept", int((o["trace_stats"][:, 3] > 0).sum()), "of", n, "; Score", o["row"].score, flush=True)
if check and os.path.exists(KMA):
    m = min(check, n)
    sub = formats.pack_ragged(reads[:m])
    fq = os.path.join(tmp, "sub.fq")
    synth.write_fastq(fq, reads[:m], prefix="r", qual=b"5")
    t0 = time.perf_counter()
    subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-Mt1", "1", "-bcNano", "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt_ref = time.perf_counter() - t0
    o = db.run_mt1(sub, 1)
    names = [f"r{i}".encode() for i in range(m)]
    db.frag_write2(os.path.join(tmp, "our.frag.gz"), sub, o["rc"], o["tmpl"], o["n_hits"], o["trace_stats"], names, order=1)
    same_frag = gzip.open(os.path.join(tmp, "our.frag.gz")).read() == gzip.open(os.path.join(tmp, "ref.frag.gz")).read()
    import golden_util
    same_fsa = golden_util.fsa_text([("genome", o["consensus"][1])]) == open(os.path.join(tmp, "ref.fsa")).read()
    line = binding.KmaHipDB.res_line("genome", o["row"], o["cover"][1], o["aln_len"][1], o["depth"][1])
    ref_res = open(os.path.join(tmp, "ref.res")).read().splitlines()
    same_res = len(ref_res) > 1 and line is not None and line.rstrip("\n") == ref_res[1]
    print(f"reference on {m} reads: {dt_ref:.2f} s = {m / dt_ref:.0f} reads/s (one thread, whole pipeline); identical: frag rows {same_frag}, consensus {same_fsa}, .res {same_res}", flush=True)
db.close()
