#!/usr/bin/env python3
"""rocprofv3 output of tools/prof_pipeline.sh -> <out>/kernel_stats.csv (the --stats summary) and <out>/pmc_traffic.json (FETCH_SIZE /
WRITE_SIZE per kernel, summed over its launches in one run of the pipeline; corrected as MI355X_MICROARCH.md prescribes for gfx950)."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out = sys.argv[1], sys.argv[2]


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


cmd = open(os.path.join(out, "cmd.txt")).read().strip()
with open(one(os.path.join(src, "stats", "**", "*kernel_stats.csv"))) as f, open(os.path.join(out, "kernel_stats.csv"), "w") as g:
    g.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}\n")
    g.write(f.read())


def counter(d, name):
    tot, cnt = {}, {}
    with open(one(os.path.join(src, d, "**", "*counter_collection.csv"))) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != name:
                continue
            k = short(row["Kernel_Name"])
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            cnt.setdefault(k, set()).add(row["Dispatch_Id"])
    return tot, {k: len(v) for k, v in cnt.items()}


fetch, nf = counter("fetch", "FETCH_SIZE")
write, nw = counter("write", "WRITE_SIZE")
res = {}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    fk, wk = fetch[k], write.get(k, 0.0)
    res[k] = {"launches": nf[k], "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "raw_bytes": (fk + wk) * 1024, "corrected_bytes": (2 * fk + wk) * 1024}
sys.path.insert(0, ROOT)
import bench  # noqa: E402
res["_kernel_src_sha256"] = bench.kernel_src_sha()
res["_note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + cmd + "; bytes per kernel summed over its "
                "launches in one run of the whole pipeline. corrected_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 as MI355X_MICROARCH.md "
                "prescribes for gfx950 (calibrated there for wide coalesced streams; raw_bytes is the lower bound)")
with open(os.path.join(out, "pmc_traffic.json"), "w") as g:
    json.dump(res, g, indent=1)
print(open(os.path.join(out, "stages.txt")).read())
with open(os.path.join(out, "kernel_stats.csv")) as f:
    for i, line in enumerate(f):
        if i < 28:
            print(line.rstrip()[:170])
for k, v in list(res.items())[:14]:
    if not k.startswith("_"):
        print(f"{k:60s} launches {v['launches']:3d} corrected {v['corrected_bytes'] / 1e9:8.3f} GB raw {v['raw_bytes'] / 1e9:8.3f} GB")
