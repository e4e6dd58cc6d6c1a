#!/usr/bin/env python3
"""The 50 k-gene database of config C5 (5 000 families x 10 variants, ~52 Mbp) on one GPU: the index from examples/kmahip_index and from
the reference's `kma index` (.length.b, .name and the bases of .seq.b identical, .comp.b the same k-mers and lists in its own order), then a read set through the reference and through
examples/kmahip_map in the three modes (-1t1, the default mode, -ipe -1t1): .res / .fsa / .frag.gz compared byte by byte.
usage (GPU box): python3 tools/big_db_check.py [reads]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from kma_amd import formats, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp(prefix="bigdb_")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
t0 = time.perf_counter()
names, seqs = synth.make_gene_db(5000, 10, 600, 1500, 0.04, seed=4321)
fa = os.path.join(tmp, "db50k.fsa")
synth.write_fasta(fa, names, seqs)
print(f"{len(seqs)} templates, {sum(len(s) for s in seqs) / 1e6:.1f} Mbp, FASTA written in {time.perf_counter() - t0:.1f} s", flush=True)
ref_db, got_db = os.path.join(tmp, "ref"), os.path.join(tmp, "got")
t0 = time.perf_counter()
subprocess.run([KMA, "index", "-i", fa, "-o", ref_db], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t1 = time.perf_counter()
subprocess.run([os.path.join(ROOT, "examples", "kmahip_index"), "-i", fa, "-o", got_db], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t2 = time.perf_counter()
same = {ext: open(ref_db + ext, "rb").read() == open(got_db + ext, "rb").read() for ext in (".length.b", ".name")}
# .comp.b: the same k-mers and as many distinct value lists (their order in the file is each builder's own; tests/test_index_gpu.py compares
# the k-mer -> template-list mapping itself on small databases; here the runs below use one index each and must agree)
ca, cb = formats.read_comp_b(got_db + ".comp.b"), formats.read_comp_b(ref_db + ".comp.b")
same[".comp.b (DB_size, k, flag, k-mers, value-list words)"] = (ca.DB_size, ca.mlen, ca.kmersize, ca.flag, ca.n, ca.prefix_len, ca.v_index) == \
    (cb.DB_size, cb.mlen, cb.kmersize, cb.flag, cb.n, cb.prefix_len, cb.v_index)
lens = formats.read_lengths(ref_db)[1:].astype(np.int64)
wa, wb = np.fromfile(got_db + ".seq.b", np.uint64), np.fromfile(ref_db + ".seq.b", np.uint64)
keep = np.ones(len(wb), bool)
ends = np.cumsum((lens >> 5) + 1) - 1
keep[ends[(lens & 31) == 0]] = False          # the word behind a template whose length is a multiple of 32: stale in the reference's file
same[".seq.b (words that carry bases)"] = len(wa) == len(wb) and bool(np.array_equal(wa[keep], wb[keep]))
print(f"kma index {t1 - t0:.1f} s, kmahip_index {t2 - t1:.1f} s (whole processes); identical: {same}", flush=True)

codes, _, _, _ = synth.make_reads(seqs, n, seed=99)
fq = os.path.join(tmp, "reads.fq")
bench.write_fastq_fixed(fq, codes)
m1, m2, _ = synth.make_pairs(seqs, n // 2, seed=98)
r1, r2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
bench.write_fastq_fixed(r1, m1)
bench.write_fastq_fixed(r2, m2)
ok = all(same.values())
for what, ref_args, got_args in (("-1t1", ["-i", fq, "-1t1"], ["-i", fq, "-1t1"]), ("default mode", ["-i", fq], ["-i", fq, "-chain"]),
                                 ("-ipe -apm p -1t1", ["-ipe", r1, r2, "-apm", "p", "-1t1"], ["-ipe", r1, r2, "-apm", "p", "-1t1"])):
    t0 = time.perf_counter()
    subprocess.run([KMA] + ref_args + ["-o", os.path.join(tmp, "ref_out"), "-t_db", ref_db, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t1 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map")] + got_args + ["-o", os.path.join(tmp, "got_out"), "-t_db", got_db], stderr=subprocess.PIPE)
    t2 = time.perf_counter()
    if r.returncode:
        print(what, "kmahip_map failed:", r.stderr.decode().strip().splitlines()[-1:], flush=True)
        ok = False
        continue
    a = [open(os.path.join(tmp, f"{x}_out.res"), "rb").read() for x in ("ref", "got")]
    b = [open(os.path.join(tmp, f"{x}_out.fsa"), "rb").read() for x in ("ref", "got")]
    c = [gzip.open(os.path.join(tmp, f"{x}_out.frag.gz")).read() for x in ("ref", "got")]
    res = (a[0] == a[1], b[0] == b[1], c[0] == c[1])
    ok = ok and all(res)
    print(f"{what}: reference {t1 - t0:.1f} s, kmahip_map {t2 - t1:.2f} s; .res {res[0]} ({a[0].count(bytes([10])) - 1} rows) .fsa {res[1]} .frag.gz {res[2]} ({c[0].count(bytes([10]))} rows)", flush=True)
print("ALL IDENTICAL" if ok else "DIFFERENCES", flush=True)
