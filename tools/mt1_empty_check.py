#!/usr/bin/env python3
"""`-Mt1 1` on inputs without a usable record (an empty file, one read that is too short, reads that map nowhere): the reference, the batched
session and the one-call path of examples/kmahip_map. usage (GPU box): python3 tools/mt1_empty_check.py"""
import gzip, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from kma_amd import synth
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(1)
g = rng.integers(0, 4, 20000, dtype=np.uint8)
prefix = os.path.join(tmp, "db")
synth.write_fasta(prefix + ".fsa", ["g"], [g])
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
cases = {"empty": [], "short": [g[:10]], "nowhere": [rng.integers(0, 4, 3000, dtype=np.uint8) for _ in range(5)]}
ok = True
for name, reads in cases.items():
    fq = os.path.join(tmp, name + ".fq")
    if reads:
        synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    else:
        open(fq, "w").close()
    r = subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-Mt1", "1", "-bcNano", "-t", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for env in ({}, {"KMAHIP_MAP_ONE_BATCH": "1"}):
        for e in ("res", "fsa", "frag.gz"):
            if os.path.exists(os.path.join(tmp, "got." + e)):
                os.unlink(os.path.join(tmp, "got." + e))
        q = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-Mt1", "1", "-bcNano"], stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        same = []
        for e in ("res", "fsa"):
            a, b = os.path.join(tmp, "ref." + e), os.path.join(tmp, "got." + e)
            same.append(os.path.exists(a) == os.path.exists(b) and (not os.path.exists(a) or open(a, "rb").read() == open(b, "rb").read()))
        a, b = os.path.join(tmp, "ref.frag.gz"), os.path.join(tmp, "got.frag.gz")
        same.append(os.path.exists(a) == os.path.exists(b) and (not os.path.exists(a) or gzip.open(a).read() == gzip.open(b).read()))
        print(name, env or "session", "reference rc", r.returncode, "kmahip_map rc", q.returncode, "identical", same, flush=True)
        ok = ok and all(same) and q.returncode == 0
print("ALL IDENTICAL" if ok else "DIFFERENCES")
