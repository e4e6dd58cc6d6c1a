#!/usr/bin/env python3
"""Long reads in the DEFAULT mode (no -1t1, no -Mt1): ONT-like reads against a gene database and against one genome, the way
`kma -i ont.fq -t_db db -bcNano` is commonly run. examples/kmahip_map -chain -bcNano against the reference on the same reads.
usage (GPU box): python3 tools/long_default_time.py [reads [read length]]"""
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from kma_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
tmp = tempfile.mkdtemp(prefix="longdef_")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
rng = np.random.default_rng(4)
genome = rng.integers(0, 4, 2_000_000, dtype=np.uint8)
prefix = os.path.join(tmp, "g")
synth.write_fasta(prefix + ".fsa", ["genome2Mb"], [genome])
subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
reads = synth.make_long_reads(genome, n, read_len=L, seed=8)
fq = os.path.join(tmp, "ont.fq")
synth.write_fastq(fq, reads, prefix="r", qual=b"5")
if os.environ.get("LONG_AB"):
    # (no reference run: the long-read route of stage 3a against the lane-per-task kernel, files compared with each other)
    out = {}
    for name, env in (("routed", {}), ("lanes", {"KMAHIP_ALIGN_LONG": "0"})):
        for rep in range(2):
            time.sleep(2)
            t1 = time.perf_counter()
            r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, name), "-chain", "-bcNano"], stderr=subprocess.PIPE,
                               env=dict(os.environ, **env))
            t2 = time.perf_counter()
            print(f"{n} x {L} nt, default mode -bcNano, stage 3a {name}: {t2 - t1:.2f} s (rc {r.returncode}) | {r.stderr.decode().strip().splitlines()[-1][:330]}", flush=True)
            for line in r.stderr.decode().splitlines():
                if "stage 3a" in line:
                    print("   ", line)
        out[name] = [open(os.path.join(tmp, f"{name}.{e}"), "rb").read() for e in ("res", "fsa")] + [gzip.open(os.path.join(tmp, f"{name}.frag.gz")).read()]
    print("files identical:", [a == b for a, b in zip(out["routed"], out["lanes"])], "fragment rows:", out["routed"][2].count(b"\n"))
    sys.exit(0)
for extra in ([], ["-bcNano"]):
    t0 = time.perf_counter()
    subprocess.run([KMA, "-i", fq, "-o", os.path.join(tmp, "ref"), "-t_db", prefix, "-t", "1"] + extra, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t1 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "got"), "-chain"] + extra, stderr=subprocess.PIPE)
    t2 = time.perf_counter()
    if r.returncode:
        print(extra, "kmahip_map failed:", r.stderr.decode().strip().splitlines()[-1:], flush=True)
        continue
    same = [open(os.path.join(tmp, f"ref.{e}"), "rb").read() == open(os.path.join(tmp, f"got.{e}"), "rb").read() for e in ("res", "fsa")]
    same.append(gzip.open(os.path.join(tmp, "ref.frag.gz")).read() == gzip.open(os.path.join(tmp, "got.frag.gz")).read())
    print(f"{n} x {L} nt, default mode {extra}: reference -t 1 {t1 - t0:.1f} s, kmahip_map {t2 - t1:.2f} s; .res / .fsa / .frag.gz identical: {same} | "
          f"{r.stderr.decode().strip().splitlines()[-1][:300]}" + ("\n" + r.stderr.decode() if os.environ.get("KMAHIP_DEBUG_TIMING") else ""), flush=True)
