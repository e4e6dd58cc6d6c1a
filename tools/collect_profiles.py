#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/<tag>/...) into the tracked
summaries under profiles/:  <round>_final_kernel_stats.csv, <round>_pmc_traffic.json,
<round>_bench_final.json.

    python tools/collect_profiles.py gpurun_out/r1f r1
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("scan_prefilter_kernel", "scan_se_kernel", "seed_tasks_kernel", "align_fast_kernel", "align_tasks_kernel", "dp_queue_kernel", "pend_finish_kernel", "scan_dense_kernel",
           "task_map_kernel", "reduce_reads_kernel")


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def counter_sum(d, counter):
    """Sum of `counter` per kernel name over all dispatches, and dispatch count."""
    tot, cnt = {}, {}
    with open(one(os.path.join(d, "**", "*_counter_collection.csv"))) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] = tot.get(name, 0.0) + float(row["Counter_Value"])
            cnt.setdefault(name, set()).add(row["Dispatch_Id"])
    return tot, {k: len(v) for k, v in cnt.items()}


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    cmd = open(os.path.join(src, "cmd.txt")).read().strip() if os.path.exists(os.path.join(src, "cmd.txt")) else ""

    stats = one(os.path.join(src, "stats", "**", "*_kernel_stats.csv"))
    with open(stats) as f, open(os.path.join(out, f"{rnd}_final_kernel_stats.csv"), "w") as g:
        g.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}\n")
        g.write(f.read())

    fetch, nf = counter_sum(os.path.join(src, "fetch"), "FETCH_SIZE")
    write, nw = counter_sum(os.path.join(src, "write"), "WRITE_SIZE")
    res = {}
    for name in fetch:
        short = next((k for k in KERNELS if k in name), None)
        if short == "dp_queue_kernel":
            short = "dp_queue_kernel" + name[name.index("<"):name.index(">") + 1]
        if short == "scan_se_kernel" and ", 64>" in name:
            short = "scan_se_kernel_tier2"                                # the 64-slot second tier (usually next to nothing to do)
        if short is None or name not in write or "<true" in name:     # <true...> = the stats-counting launches
            continue
        fk = fetch[name] / nf[name]
        wk = write[name] / nw[name]
        res[short] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                      "raw_bytes": (fk + wk) * 1024, "corrected_bytes": (2 * fk + wk) * 1024,
                      "launches": nf[name]}
    sys.path.insert(0, ROOT)
    import bench
    res["_kernel_src_sha256"] = bench.kernel_src_sha()
    res["_note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + cmd +
                    "; bytes per launch (mean over launches). corrected_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 as "
                    "MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 128-B requests at 64 B; calibrated "
                    "there for wide coalesced streams only -- our 32-B random gathers are uncalibrated, raw_bytes is "
                    "the lower bound).")
    with open(os.path.join(out, f"{rnd}_pmc_traffic.json"), "w") as g:
        json.dump(res, g, indent=1)

    with open(os.path.join(src, "bench.json")) as f:
        line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    with open(os.path.join(out, f"{rnd}_bench_final.json"), "w") as g:
        g.write(line + "\n")
    print(json.dumps(res, indent=1))
    print(line)


if __name__ == "__main__":
    main()
