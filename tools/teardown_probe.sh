#!/bin/bash
# Run ON THE GPU BOX: where does kmahip_map's process lifetime go beyond its own wall clock? Whole-process time of runs stopped after
# open (HIP start-up + index), and of complete runs with different scratch sizes.
set -e
python3 - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from kma_amd import formats, synth
import numpy as np
tmp = tempfile.mkdtemp()
names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
prefix = os.path.join(tmp, "db")
formats.write_index(prefix, names, seqs)
sys.path.insert(0, os.getcwd())
import bench
n = 10_000_000
codes, *_ = synth.make_reads(seqs, 1_000_000, seed=5)
fq = os.path.join(tmp, "r.fq")
with open(fq, "wb") as f:
    for i in range(n // 1_000_000):
        bench.write_fastq_fixed(os.path.join(tmp, "p.fq"), codes)
        f.write(open(os.path.join(tmp, "p.fq"), "rb").read())
subprocess.check_call(["make", "-C", "examples"], stdout=subprocess.DEVNULL)
def run(env, args=()):
    t0 = time.perf_counter()
    r = subprocess.run(["examples/kmahip_map", "-i", fq, "-t_db", prefix, "-o", os.path.join(tmp, "o"), "-1t1", *args], env=dict(os.environ, **env), stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    last = r.stderr.decode().strip().splitlines()[-1] if r.stderr else ""
    return dt, last
for label, env, args in (("whole run, teardown timed", {"KMAHIP_MAP_TEARDOWN": "1"}, ()), ("stop after open", {"KMAHIP_MAP_STOP": "open", "KMAHIP_MAP_ONE_BATCH": "1"}, ()), ("whole run", {}, ()), ("whole run, -nf -na", {}, ("-nf", "-na")),
                         ("whole run, early return", {"KMAHIP_MAP_EARLY_RETURN": "1"}, ())):
    for rep in range(2):
        dt, last = run(env, args)
        print(f"{label}: {dt:.3f} s | {last[:200]}", flush=True)
        time.sleep(1.0)
PY
