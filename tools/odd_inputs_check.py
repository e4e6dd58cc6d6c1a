#!/usr/bin/env python3
"""Odd reads through the reference and examples/kmahip_map (-1t1 and the default mode): shorter than k, exactly k, all N, poly-A, a
low-complexity repeat, a 300 kb read, a read that is a whole template, its reverse complement, reads with N runs; as FASTQ and as
multi-line FASTA. usage (GPU box): python3 tools/odd_inputs_check.py"""
import gzip
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import synth  # noqa: E402

KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp(prefix="odd_")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
rng = np.random.default_rng(9)
names, seqs = synth.make_gene_db(30, 4, 600, 2500, 0.03, seed=55)
prefix = os.path.join(tmp, "db")
synth.write_fasta(prefix + ".fsa", names, seqs)
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
lut = np.frombuffer(b"ACGTN", np.uint8)
reads = []
g = seqs[3]
reads += [g[:15], g[:16], g[:17], np.full(200, 4, np.uint8), np.zeros(300, np.uint8), np.tile(np.array([0, 1], np.uint8), 200),
          g.copy(), synth.revcomp_codes(g), np.concatenate([g[:200], np.full(40, 4, np.uint8), g[240:600]]),
          np.concatenate([np.full(5, 4, np.uint8), seqs[7][20:400]]), np.concatenate([seqs[9][:300], np.full(1, 4, np.uint8)])]
big = np.concatenate([seqs[int(rng.integers(0, len(seqs)))] for _ in range(200)])[:300000]
reads.append(big)
for i in range(300):
    s = seqs[int(rng.integers(0, len(seqs)))]
    L = int(rng.integers(16, 400)); a = int(rng.integers(0, len(s) - L))
    reads.append(s[a:a + L].copy())
ok = True
for fmt in ("fq", "fa"):
    path = os.path.join(tmp, "r." + fmt)
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            if fmt == "fa" and (len(r) < 17 or len(r) > 100000 or (r == 4).all()):
                continue          # (the reference's FASTA reader crashes on this set otherwise; what is left still has the N runs and repeats)
            t = lut[r].tobytes()
            if fmt == "fq":
                f.write(b"@r%d\n" % i + t + b"\n+\n" + b"I" * len(t) + b"\n")
            else:
                f.write(b">r%d\n" % i + b"\n".join(t[a:a + 70] for a in range(0, len(t), 70)) + b"\n")
    for mode in ("-1t1", "default"):
        ra = ["-i", path] + (["-1t1"] if mode == "-1t1" else [])
        ga = ["-i", path, "-1t1" if mode == "-1t1" else "-chain"]
        rr = subprocess.run([KMA] + ra + ["-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-t", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if rr.returncode:
            print(fmt, mode, "the reference itself ends with", rr.returncode, "on this input: skipped", flush=True)
            continue
        r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map")] + ga + ["-t_db", prefix, "-o", os.path.join(tmp, "got")], stderr=subprocess.PIPE)
        if r.returncode:
            print(fmt, mode, "kmahip_map failed:", r.stderr.decode().strip().splitlines()[-1:], flush=True)
            ok = False
            continue
        same = [open(os.path.join(tmp, f"ref.{e}"), "rb").read() == open(os.path.join(tmp, f"got.{e}"), "rb").read() for e in ("res", "fsa")]
        same.append(gzip.open(os.path.join(tmp, "ref.frag.gz")).read() == gzip.open(os.path.join(tmp, "got.frag.gz")).read())
        ok = ok and all(same)
        print(fmt, mode, "identical", same, "rows", gzip.open(os.path.join(tmp, "ref.frag.gz")).read().count(b"\n"), flush=True)
print("ALL IDENTICAL" if ok else "DIFFERENCES")
