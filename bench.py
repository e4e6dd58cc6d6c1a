#!/usr/bin/env python3
"""Benchmark of the KMA mapping hot path on MI355X (driver contract: see the task brief).

One "step" = one pass of the HIP hot path over one batch of synthetic reads that
is already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]:
10 M x 150 bp single-end reads against a 5 k-gene database, `-1t1`.  With N > 1
every rank maps its own 10 M-read shard (weak scaling); reads are independent,
so there is no data-path collective inside stage 2.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU per step")
    ap.add_argument("--families", type=int, default=1000, help="gene families (x5 variants = genes)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="reads timed on the host CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def write_fastq_fixed(path, codes):
    """Vectorised FASTQ writer: fixed-width names so every record has one size."""
    n, L = codes.shape
    lut = np.frombuffer(b"ACGT", np.uint8)
    name_w = 9
    rec = 1 + name_w + 1 + L + 3 + L + 1
    buf = np.empty((n, rec), np.uint8)
    buf[:, 0] = ord("@")
    idx = np.arange(n)
    buf[:, 1] = ord("r")
    for d in range(name_w - 1):
        buf[:, 1 + name_w - 1 - d] = ord("0") + (idx // 10 ** d) % 10
    o = 1 + name_w
    buf[:, o] = ord("\n"); o += 1
    buf[:, o:o + L] = lut[codes]; o += L
    buf[:, o:o + 3] = np.frombuffer(b"\n+\n", np.uint8); o += 3
    buf[:, o:o + L] = ord("I"); o += L
    buf[:, o] = ord("\n")
    with open(path, "wb") as f:
        f.write(buf.tobytes())


def cpu_baseline(prefix, codes, tmp):
    """Reference KMA (oracle/_ref/kma, stage 1 + stage 2 via its -s2 tap) or, if the
    binary is absent, the C oracle port, timed on a bounded sample of the same reads."""
    n = len(codes)
    ref = os.path.join(ROOT, "oracle", "_ref", "kma")
    if os.path.exists(ref):
        fq = os.path.join(tmp, "sample.fq")
        write_fastq_fixed(fq, codes)
        cmd = [ref, "-i", fq, "-o", os.path.join(tmp, "cpu"), "-t_db", prefix, "-1t1", "-t", "1", "-s2"]
        t0 = time.time()
        with open(os.devnull, "wb") as dn:
            subprocess.run(cmd, stdout=dn, stderr=dn, check=True)
        dt = time.time() - t0
        return dict(value=n / dt, unit="reads/s", cores=2, kind="reference",
                    sample=f"{n} of the step's reads; reference kma -1t1 -t 1 -s2 (stage-1 FASTQ parse thread + "
                           f"one stage-2 scan thread), {dt:.1f} s wall")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from kma_amd import formats
    batch = formats.pack_fixed(codes)
    odb = oracle.OracleDB(prefix)
    t0 = time.time()
    odb.scan_se(batch)
    dt = time.time() - t0
    return dict(value=n / dt, unit="reads/s", cores=1, kind="port",
                sample=f"{n} of the step's reads; oracle/scan.c scalar port, {dt:.1f} s")


def parity_sample(prefix, codes, got):
    """Checker leg: the first reads of the step vs the CPU oracle (bit-exact)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from kma_amd import formats
    batch = formats.pack_fixed(codes)
    e = oracle.OracleDB(prefix).scan_se(batch)
    return all(np.array_equal(a, b) for a, b in zip(e, got))


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from kma_amd import binding, formats, synth, synth_dev
    tmp = tempfile.mkdtemp(prefix=f"kmabench{rank}_")
    try:
        names, seqs = synth.make_gene_db(a.families, 5, 600, 1500, 0.04, seed=12345)
        prefix = os.path.join(tmp, "db5k")
        formats.write_index(prefix, names, seqs)
        db = binding.KmaHipDB(prefix, device=local)
        n = a.reads
        keep = min(n, a.cpu_sample) if rank == 0 else 0
        rd = synth_dev.make_packed_reads(seqs, n, seed=1000 + rank, device=dev, keep_codes=keep)
        rc_flag = torch.empty(n, dtype=torch.int32, device=dev)
        flag = torch.empty(n, dtype=torch.int32, device=dev)
        T_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        T = torch.empty(8 * n, dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def step():
            db.scan_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], rc_flag, flag, T_off, T,
                           stream=stream)

        def fence():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(a.warmup):
            step()
        db.status(stream)
        db.set_timing(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        kern_ms, launches = db.get_timing()
        db.set_timing(False)
        db.status(stream)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

        # algorithmic work of one launch (separate, untimed, counter-enabled launch)
        db.set_stats(True)
        step()
        st = db.get_stats(stream)
        db.set_stats(False)
        mapped = int((T_off[1:] > T_off[:-1]).sum().item())
        total_T = int(T_off[-1].item())
        W = (150 + 31) // 32
        # 12 B per probe (4 B bucket directory + 4 B key + 4 B value index in the
        # reference layout), 2 B per value-list element read, packed read in, S2 fields out
        alg_bytes = 12 * st.probes + 2 * st.value_elems + n * (8 * W + 4 + 8) + n * (4 + 4 + 8) + 4 * total_T
        kern_s = kern_ms / 1e3 / max(1, launches)
        achieved = alg_bytes / kern_s / 1e9

        out = {
            "metric": "mapped reads/sec (whole node), 10M×150bp vs 5k-gene DB, 1/2/4/8 GPU",
            "value": world * n * a.steps / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": f"{n} x 150 bp SE reads per GPU vs {5 * a.families}-gene DB (k=16), -1t1; "
                            "hot path covered: stage 2 (k-mer probe + candidate-template scoring, "
                            "save_kmers/hashMap_get); stage 3a not yet on device",
                "reads_per_gpu": n, "genes": 5 * a.families, "db_kmers": int(db.info.n_kmers),
                "probe_table_MB": round(db.info.hash_bytes / 1e6, 1),
                "mapped_fraction": mapped / n, "parallelism": f"read-shard x{world}",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": "scan_se_kernel", "kernel_ms": kern_s * 1e3,
                "algorithmic_bytes_per_launch": alg_bytes,
                "probes_per_launch": int(st.probes), "probes_per_s": st.probes / kern_s,
            },
        }
        if rank == 0 and world == 1 and not a.no_cpu and keep:
            codes = rd["codes"][:keep]
            k = min(50_000, keep)
            got = [x.cpu().numpy() for x in (rc_flag[:k], flag[:k], T_off[:k + 1])]
            got.append(T[: int(got[2][-1])].cpu().numpy())
            out["config"]["parity_first_reads_vs_oracle"] = bool(parity_sample(prefix, codes[:k], got))
            out["cpu_baseline"] = cpu_baseline(prefix, codes, tmp)
        elif rank == 0:
            out["cpu_baseline"] = None
        if rank == 0:
            print(json.dumps(out), flush=True)
        db.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
