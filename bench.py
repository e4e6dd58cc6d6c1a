#!/usr/bin/env python3
"""Benchmark of the KMA mapping hot path on MI355X (driver contract: see the task brief).

One "step" = one pass of the HIP hot path over one batch of synthetic reads that
is already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]:
10 M x 150 bp single-end reads against a 5 k-gene database, `-1t1`.  With N > 1
every rank maps its own 10 M-read shard (weak scaling, the default) or its share
of the fixed 10 M reads (`--scaling strong`; a weak run also reports the strong
figure as the extra key `strong_scaling`); reads are independent, so the only
data-path collective is the SUM of the two ConClave score vectors (RCCL).

`python bench.py --gpus N` without a launcher starts the N ranks itself (a
torch.distributed.run child, before this process touches the GPU); under a
launcher (RANK / WORLD_SIZE in the environment) it is one of the ranks.

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
    ap.add_argument("--parity-sample", type=int, default=50_000,
                    help="reads of the step re-checked against the CPU oracle (bounded by --cpu-sample; the oracle does ~0.1 M reads/s)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--hard", action="store_true", help="NOT the BASELINE workload: 2 %% unmappable reads and 2 %% reads whose last "
                    "60-120 bases are foreign (stress of the prefilter rejection and of the unaligned-end DP); for DESIGN.md only")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--c4-reads", type=int, default=1_000_000, help="reads of the extra BASELINE config C4 leg (10 kb ONT-like reads vs one 5 Mb "
                    "genome, -Mt1 1 -bcNano, whole run incl. pile-up and consensus; the config's size is 1 M); 0 = skip")
    ap.add_argument("--c4-parity", type=int, default=5000, help="reads of the C4 leg that also go through the reference binary (identical files)")
    ap.add_argument("--e2e-reads", type=int, default=10_000_000, help="reads of the file-to-file leg (FASTQ -> .res / .fsa / .frag.gz through "
                    "examples/kmahip_map, whole-process wall clock); 0 = skip")
    ap.add_argument("--e2e-sample", type=int, default=1_000_000, help="reads of it the reference binary is run on (parity of .res + its rates)")
    ap.add_argument("--pe-pairs", type=int, default=10_000_000, help="pairs of the extra BASELINE config C3 legs (resident step and file to file); 0: skip")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --reads per rank; strong: --reads in total, sharded over the ranks")
    return ap.parse_args()


def launch_ranks(a):
    """`--gpus N` given to a plain python process: start the N ranks as a child launcher. Nothing in this process has touched
    the GPU yet (no torch.cuda call, no HIP call), and the child is a new process, not an exec of this one."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def kernel_src_sha():
    """hash of the kernel sources a PMC traffic figure belongs to"""
    import hashlib
    h = hashlib.sha256()
    for f in ("scan.hip", "align.hip", "kmahip_internal.h"):
        h.update(open(os.path.join(ROOT, "kma_amd", "csrc", f), "rb").read())
    return h.hexdigest()


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
    """Reference KMA (oracle/_ref/kma) on a bounded sample of the step's reads, using the
    reference's own per-stage CPU timers (-status): stage 2 "ankering" + stage 3a "KMA mapping".
    Falls back to the C oracle port when the binary is absent."""
    import re
    n = len(codes)
    ref = os.path.join(ROOT, "oracle", "_ref", "kma")
    if os.path.exists(ref):
        fq = os.path.join(tmp, "sample.fq")
        write_fastq_fixed(fq, codes)
        cmd = [ref, "-i", fq, "-o", os.path.join(tmp, "cpu"), "-t_db", prefix, "-1t1", "-t", "1", "-status",
               "-nc", "-na", "-nf"]
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True)
        wall = time.time() - t0
        err = r.stderr.decode()
        m2 = re.search(r"ankering query:\s*([0-9.]+) s", err)
        m3 = re.search(r"KMA mapping time\s*([0-9.]+) s", err)
        if m2 and m3:
            s2, s3 = float(m2.group(1)), float(m3.group(1))
            return dict(value=n / (s2 + s3), unit="reads/s", cores=1, kind="reference",
                        sample=f"{n} of the step's reads; reference kma -1t1 -t 1 -status: its own stage timers give "
                               f"stage 2 (ankering) {s2:.2f} s + stage 3a (mapping) {s3:.2f} s of CPU time, one thread "
                               f"each (whole pipeline incl. parse/ConClave/assembly: {wall:.1f} s wall)",
                        stage2_s=s2, stage3a_s=s3, whole_pipeline_wall_s=wall)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from kma_amd import formats
    batch = formats.pack_fixed(codes)
    odb = oracle.OracleDB(prefix)
    t0 = time.time()
    e = odb.scan_se(batch)
    odb.align_se(batch, *e)
    dt = time.time() - t0
    return dict(value=n / dt, unit="reads/s", cores=1, kind="port",
                sample=f"{n} of the step's reads; oracle/scan.c + oracle/align.c scalar port, {dt:.1f} s")


VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9     # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 32 int32 lanes/clk (wave64 in 2 passes) x 2.4 GHz = 78.6 T lane-ops/s
DP_OPS_PER_CELL = 14                     # ~14 integer operations per affine-gap DP cell


def c4_leg(tmp, n_reads, device, n_parity=5000):
    """BASELINE config C4 beside the headline step (extra key, never `value`): n_reads ONT-like reads of 10 kb (4 % substitutions,
    3 % deletions, 3 % insertions) against ONE random 5 Mb genome, `-Mt1 1 -bcNano`: per read strand choice (anker_rc), chaining and
    traceback joins (longtrace.hip), pile-up in stream order and nanoCaller consensus, through kmahip_run_mt1 (host buffers in, so the
    upload is inside). The reference binary runs the first 1000 reads on one host core for its rate and for the parity of
    `.res` / consensus / fragment rows."""
    import gzip
    from kma_amd import binding, formats, synth, synth_dev
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    G, L = 5_000_000, 10_000
    rng = np.random.default_rng(4)
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    prefix = os.path.join(tmp, "g5mb")
    ref = os.path.join(ROOT, "oracle", "_ref", "kma")
    if os.path.exists(ref):
        synth.write_fasta(prefix + ".fsa", ["genome5Mb"], [genome])
        subprocess.run([ref, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        formats.write_index(prefix, ["genome5Mb"], [genome])
    t0 = time.perf_counter()
    m = min(n_parity, n_reads)
    rd = synth_dev.make_long_reads_packed(genome, n_reads, read_len=L, seed=8, device=f"cuda:{device}", keep_codes=m)
    reads = rd["codes"]
    b = formats.ReadBatch(rd["seq"], rd["seq_off"], rd["length"], rd["N"][:0], rd["N_off"])
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    db = binding.KmaHipDB(prefix, device=device)
    t_open = time.perf_counter() - t0
    try:
        # the warm-up is the call itself, once (as the headline step has its warm-up steps): scratch of every pass at its final size, both
        # sets of the traceback's pools, the output arrays' pages touched -- a first call of this size took 190-220 ms per 200 k reads for the
        # trace stage where every later one takes 155-158 (tools/mt1_repeat.py)
        db.run_mt1(b, 1, consensus=False)
        t0 = time.perf_counter()
        o = db.run_mt1(b, 1, consensus=False)
        dt = time.perf_counter() - t0
        st = db.get_trace_stats()
        bases = int(b.length.sum())
        out = {"workload": f"{n_reads} x 10 kb ONT-like reads (4/3/3 % sub/del/ins) vs one 5 Mb genome, -Mt1 1 -bcNano: strand choice, chain, "
                           "traceback joins, stream-order pile-up, nanoCaller consensus; kmahip_run_mt1, host buffers in",
               "reads": n_reads, "reads_per_s": n_reads / dt, "gbases_per_s": bases / dt / 1e9, "call_ms": dt * 1e3, "db_open_s": round(t_open, 2),
               "stage_ms": {k: round(v, 2) for k, v in zip(("upload", "-", "figures", "trace", "pileup+consensus", "copies"), o["ms"])},
               "kept_reads": int((o["trace_stats"][:, 3] > 0).sum()), "dp_problems": int(st.problems), "dp_cells": int(st.dp_cells),
               "mems_chained": int(st.mems), "trace_stage_GCUPS": st.dp_cells / (o["ms"][3] / 1e3) / 1e9 if o["ms"][3] else None,
               "mean_depth": bases / G, "reads_generated_on_device_s": round(t_gen, 1)}
        # the contract's roofline view of this path: what it must move through HBM at the least (12 B per position-index
        # lookup, one per base and strand; the packed reads; one move byte written and read per DP cell that leaves LDS -- here none
        # of the common classes does; the runs and figures out) against what the stage takes. The DP lives in LDS and on the VALUs.
        alg = 2 * bases * 12 + bases // 4 + int(st.mems) * 40 + n_reads * 64
        out["roofline"] = {"bound": "valu + lds (integer DP, move matrices in LDS); HBM carries the index lookups", "algorithmic_bytes": alg,
                           "achieved": alg / (o["ms"][3] / 1e3) / 1e9 if o["ms"][3] else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": alg / (o["ms"][3] / 1e3) / 1e9 / HBM_PEAK_GBS if o["ms"][3] else None,
                           "GCUPS": out["trace_stage_GCUPS"]}
        # ... and the bound that does apply to the DP: integer VALU issue. peak = lane-ops/s of the chip / operations per cell
        peak_gcups = VALU_LANE_OPS / DP_OPS_PER_CELL / 1e9
        out["valu_roofline"] = {"bound": "valu (int32)", "achieved": out["trace_stage_GCUPS"], "peak": peak_gcups, "unit": "GCUPS",
                                "frac": out["trace_stage_GCUPS"] / peak_gcups if out["trace_stage_GCUPS"] else None,
                                "lane_ops_per_s": VALU_LANE_OPS, "ops_per_cell": DP_OPS_PER_CELL,
                                "note": "DP cells of the trace stage / the whole trace stage's time (seeding, chaining and the walks included)"}
        if os.path.exists(ref) and m:
            import golden_util
            sub = formats.pack_ragged(reads[:m])
            fq = os.path.join(tmp, "c4sub.fq")
            synth.write_fastq(fq, reads[:m], prefix="r", qual=b"5")
            t0 = time.perf_counter()
            subprocess.run([ref, "-i", fq, "-o", os.path.join(tmp, "c4ref"), "-t_db", prefix, "-Mt1", "1", "-bcNano", "-t", "1"], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt_ref = time.perf_counter() - t0
            os_ = db.run_mt1(sub, 1)
            db.frag_write2(os.path.join(tmp, "c4our.frag.gz"), sub, os_["rc"], os_["tmpl"], os_["n_hits"], os_["trace_stats"], [f"r{i}".encode() for i in range(m)], order=1)
            same = gzip.open(os.path.join(tmp, "c4our.frag.gz")).read() == gzip.open(os.path.join(tmp, "c4ref.frag.gz")).read()
            same = same and golden_util.fsa_text([("genome5Mb", os_["consensus"][1])]) == open(os.path.join(tmp, "c4ref.fsa")).read()
            line = binding.KmaHipDB.res_line("genome5Mb", os_["row"], os_["cover"][1], os_["aln_len"][1], os_["depth"][1])
            ref_res = open(os.path.join(tmp, "c4ref.res")).read().splitlines()
            same = same and len(ref_res) > 1 and line is not None and line.rstrip("\n") == ref_res[1]
            out["cpu_reference"] = {"reads": m, "reads_per_s": m / dt_ref, "cores": 1, "what": "oracle/_ref/kma -Mt1 1 -bcNano -t 1, whole process wall clock"}
            out["parity_subset_identical"] = bool(same)
        return out
    finally:
        db.close()


def capture_stderr(fn):
    """fn() with the process's fd 2 in a temporary file (the library prints its timing lines there): (result, text)"""
    import tempfile
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile(mode="w+b") as tf:
        os.dup2(tf.fileno(), 2)
        try:
            res = fn()
        finally:
            sys.stderr.flush()
            os.dup2(saved, 2)
            os.close(saved)
        tf.seek(0)
        text = tf.read().decode(errors="replace")
    return res, text


def chain_stage2_leg(db, pb, tmp, device):
    """Stage 2 of the reference's DEFAULT mode (no -1t1: save_kmers_chain) on its own, beside the headline step (extra key, never `value`):
    (a) the step's 10 M short reads through kmahip_scan_chain -- the library's own figure for its two kernels (anchors + chaining, the
    chunks overlapped; KMAHIP_CHAIN_TIMING); (b) 20 000 ONT-like reads of 10 kb against one 2 Mb genome -- the wavefront-per-read route."""
    import re
    from kma_amd import binding, formats, synth_dev
    out = {}
    os.environ["KMAHIP_CHAIN_TIMING"] = "1"
    try:
        db.scan_chain(formats.ReadBatch(pb.seq[:pb.seq_off[1000]], pb.seq_off[:1001], pb.length[:1000], pb.N, pb.N_off[:1001]))      # first launches, streams
        _, text = capture_stderr(lambda: db.scan_chain(pb))
        m = re.findall(r"fast route, (\d+) reads in (\d+) chunks.*?: ([0-9.]+) ms", text)
        if m:
            n_s, chunks, ms = int(m[-1][0]), int(m[-1][1]), float(m[-1][2])
            out["short_reads"] = {"reads": n_s, "chunks": chunks, "ms": ms, "reads_per_s": n_s / (ms / 1e3),
                                  "what": "prefilter + chain_anchor_kernel + chain_fast_kernel over the step's reads, device time of the route"}
        rng = np.random.default_rng(4)
        genome = rng.integers(0, 4, 2_000_000, dtype=np.uint8)
        prefix = os.path.join(tmp, "g2mb")
        formats.write_index(prefix, ["genome2Mb"], [genome])
        n_l = 20000
        rd = synth_dev.make_long_reads_packed(genome, n_l, read_len=10000, seed=8, device=f"cuda:{device}", keep_codes=0)
        lb = formats.ReadBatch(rd["seq"], rd["seq_off"], rd["length"], rd["N"][:0], rd["N_off"])
        dbl = binding.KmaHipDB(prefix, device=device)
        try:
            dbl.scan_chain(formats.ReadBatch(lb.seq[:lb.seq_off[64]], lb.seq_off[:65], lb.length[:64], lb.N, lb.N_off[:65]))
            _, text = capture_stderr(lambda: dbl.scan_chain(lb))
        finally:
            dbl.close()
        m = re.findall(r"long-read route, (\d+) reads in (\d+) chunk\(s\): ([0-9.]+) ms", text)
        if m:
            n_r, chunks, ms = int(m[-1][0]), int(m[-1][1]), float(m[-1][2])
            out["long_reads"] = {"reads": n_r, "read_len": 10000, "chunks": chunks, "ms": ms, "reads_per_s": n_r / (ms / 1e3),
                                 "what": "chain_long_anchor_kernel (a wavefront per read and strand) + chain_long_tail_kernel, device time of the route"}
    finally:
        os.environ.pop("KMAHIP_CHAIN_TIMING", None)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def e2e_leg(tmp, prefix, seqs, n, sample, log=None, pe_pairs=1_000_000, device="cuda:0"):
    """File to file: a FASTQ of n reads -> .res / .fsa / .frag.gz through the C host program (examples/kmahip_map): whole-process
    wall clock including HIP start-up, kmahip_db_open, ingest, the device run and the three writers, plain and gzip-compressed
    input. The compiled reference on the first `sample` reads of the same file: -t 1, -t nproc, and independent -t 1 processes
    over equal shards (16 = the CPU share of one GPU on the box, and nproc); its .res must be the one kmahip_map writes for
    that sample."""
    import re
    from kma_amd import synth
    say = log or (lambda *_: None)
    kma = os.path.join(ROOT, "oracle", "_ref", "kma")
    mapper = os.path.join(ROOT, "examples", "kmahip_map")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    fq = os.path.join(tmp, "e2e.fq")
    t0 = time.perf_counter()
    with open(fq, "wb") as f:
        for a in range(0, n, 2_000_000):
            m = min(2_000_000, n - a)
            codes, _, _, _ = synth.make_reads(seqs, m, seed=1000 + a)
            part = os.path.join(tmp, "part.fq")
            write_fastq_fixed(part, codes)          # (names restart per part; nobody minds, and both sides see the same file)
            with open(part, "rb") as g:
                shutil.copyfileobj(g, f, 1 << 24)
            os.unlink(part)
    rec = os.path.getsize(fq) // n
    say(f"e2e: FASTQ of {n} reads, {os.path.getsize(fq) / 1e9:.2f} GB, written in {time.perf_counter() - t0:.1f} s")
    nproc = len(os.sched_getaffinity(0))
    # the same file gzip-compressed (level 1), made by 16 gzip processes over slices (a multi-member .gz)
    per = (n + 15) // 16
    procs = []
    with open(fq, "rb") as f:
        for i in range(16):
            sl = os.path.join(tmp, f"slice{i:02d}.fq")
            with open(sl, "wb") as g:
                g.write(f.read(rec * per))
            procs.append(subprocess.Popen(["gzip", "-1", sl]))
    for p_ in procs:
        p_.wait()
    gz = fq + ".gz"
    with open(gz, "wb") as f:
        for i in range(16):
            with open(os.path.join(tmp, f"slice{i:02d}.fq.gz"), "rb") as g:
                shutil.copyfileobj(g, f, 1 << 24)
            os.unlink(os.path.join(tmp, f"slice{i:02d}.fq.gz"))

    def run_map(inp, outp, env=None):
        t0 = time.perf_counter()
        r = subprocess.run([mapper, "-i", inp, "-t_db", prefix, "-o", outp, "-1t1"], stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
        dt = time.perf_counter() - t0
        err = r.stderr.decode().strip().splitlines()
        if r.returncode:
            raise RuntimeError(f"kmahip_map failed ({r.returncode}): {err[-1] if err else ''}")
        for line in err[:-1]:
            if any(k in line for k in ("write_rows", "frag_write", "db_open", "ingest:")):
                say("    " + line)
        return dt, err[-1] if err else ""

    out = {"reads": n, "fastq_GB": round(os.path.getsize(fq) / 1e9, 2), "fastq_gz_GB": round(os.path.getsize(gz) / 1e9, 2), "unit": "reads/s",
           "host_threads": min(16, nproc),
           "what": "examples/kmahip_map -i <fastq> -t_db <index> -o <out> -1t1 as ONE process: from its start until it is gone (HIP start-up, "
                   "kmahip_db_open, ingest, the batched session: upload, stages 2 + 3a, ConClave, traceback, pile-up, consensus; .res, .fsa, "
                   ".frag.gz; and the teardown of its mappings and of the device context) -- the wall the reference's is compared with. "
                   "outputs_closed: the same run with KMAHIP_MAP_EARLY_RETURN=1, where the command returns once every output is closed and a "
                   "child finishes the teardown unwaited-for; reported beside it, never used for a ratio"}
    got = os.path.join(tmp, "e2e_got")
    walls = [run_map(fq, got) for _ in range(2)]
    best = min(walls)
    out["plain"] = {"wall_s": round(best[0], 3), "reads_per_s": n / best[0], "runs_s": [round(w[0], 3) for w in walls]}
    m = re.search(r"wall: (.*?) \|", best[1])
    if m:
        out["plain"]["breakdown"] = m.group(1)
    say(f"e2e: plain {best[0]:.2f} s | {best[1]}")
    time.sleep(1.0)
    we = [run_map(fq, got + "_early", env={"KMAHIP_MAP_EARLY_RETURN": "1"})[0]]
    time.sleep(1.5)          # (the orphan of that run is still giving the device back)
    we.append(run_map(fq, got + "_early", env={"KMAHIP_MAP_EARLY_RETURN": "1"})[0])
    time.sleep(1.5)
    out["outputs_closed"] = {"wall_s": round(min(we), 3), "reads_per_s": n / min(we), "runs_s": [round(w, 3) for w in we],
                             "teardown_s": round(best[0] - min(we), 3)}
    say(f"e2e: outputs closed after {min(we):.2f} s (early return); teardown {best[0] - min(we):.2f} s")
    wz = run_map(gz, got + "_gz")
    out["gz"] = {"wall_s": round(wz[0], 3), "reads_per_s": n / wz[0]}
    m = re.search(r"wall: (.*?) \|", wz[1])
    if m:
        out["gz"]["breakdown"] = m.group(1)
    out["gz"]["res_equals_plain"] = open(got + ".res", "rb").read() == open(got + "_gz.res", "rb").read()
    say(f"e2e: gz {wz[0]:.2f} s | {wz[1]}")
    ref = None
    if os.path.exists(kma) and sample:
        m_ = min(sample, n)
        sfq = os.path.join(tmp, "e2e_sample.fq")
        with open(fq, "rb") as f, open(sfq, "wb") as g:
            g.write(f.read(rec * m_))
        ws = run_map(sfq, got + "_sample")

        def run_ref(threads, inp, outp):
            return subprocess.Popen([kma, "-i", inp, "-o", outp, "-t_db", prefix, "-1t1", "-t", str(threads)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

        def shards(k):
            per_ = (m_ + k - 1) // k
            paths = []
            with open(sfq, "rb") as f:
                for i in range(k):
                    p_ = os.path.join(tmp, f"shard{i}.fq")
                    with open(p_, "wb") as g:
                        g.write(f.read(rec * per_))
                    paths.append(p_)
            t0 = time.perf_counter()
            ps = [run_ref(1, p_, os.path.join(tmp, f"shard_out{i}")) for i, p_ in enumerate(paths)]
            bad = sum(1 for q in ps if q.wait() != 0)
            dt = time.perf_counter() - t0
            for p_ in paths:
                os.unlink(p_)
            if bad:
                raise RuntimeError(f"{bad} of {k} reference processes failed")
            return dt
        t0 = time.perf_counter()
        if run_ref(1, sfq, os.path.join(tmp, "e2e_ref")).wait():
            raise RuntimeError("reference run failed")
        t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        run_ref(nproc, sfq, os.path.join(tmp, "e2e_reft")).wait()
        tn = time.perf_counter() - t0
        s16 = shards(min(16, nproc))
        sn = shards(nproc) if nproc > 16 else s16
        # the best the reference can do on this host with the WHOLE file: k independent -t 1 processes over equal byte slices of all
        # n reads (wall = the slowest), k = 16 (one GPU's CPU share), 64, 128, nproc. Slices are written once, nproc of them; a
        # process of a smaller k takes several (`-i f1 f2 ...`).
        parts = max(nproc, 16)
        per_p = (n + parts - 1) // parts
        slices = []
        with open(fq, "rb") as f:
            for i in range(parts):
                blob = f.read(rec * per_p)
                if not blob:
                    break
                sl = os.path.join(tmp, f"full{i:03d}.fq")
                with open(sl, "wb") as g:
                    g.write(blob)
                slices.append(sl)
        full = {}
        for k in sorted({min(16, nproc), min(64, nproc), min(128, nproc), nproc}):
            groups = [list(x) for x in np.array_split(np.arange(len(slices)), k) if len(x)]
            t0 = time.perf_counter()
            ps = [subprocess.Popen([kma, "-i", *[slices[j] for j in grp], "-o", os.path.join(tmp, f"full_out{i}"), "-t_db", prefix, "-1t1", "-t", "1"],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i, grp in enumerate(groups)]
            bad = sum(1 for q in ps if q.wait() != 0)
            dt_k = time.perf_counter() - t0
            if bad:
                raise RuntimeError(f"{bad} of {k} reference processes failed on the full file")
            full[str(k)] = {"processes": len(groups), "reads_per_process": n // len(groups), "wall_s": round(dt_k, 2), "reads_per_s": n / dt_k}
            say(f"e2e: reference, {len(groups)} processes x -t 1 over the whole file ({n} reads): {dt_k:.1f} s = {n / dt_k / 1e6:.2f} M reads/s")
        for sl in slices:
            os.unlink(sl)
        same = open(got + "_sample.res", "rb").read() == open(os.path.join(tmp, "e2e_ref.res"), "rb").read()
        ref = {"sample_reads": m_, "cpu_model": cpu_model(), "nproc": nproc,
               "t1": {"wall_s": round(t1, 2), "reads_per_s": m_ / t1},
               "t_nproc": {"threads": nproc, "wall_s": round(tn, 2), "reads_per_s": m_ / tn},
               "shards_16": {"processes": min(16, nproc), "wall_s": round(s16, 2), "reads_per_s": m_ / s16},
               "shards_nproc": {"processes": nproc, "wall_s": round(sn, 2), "reads_per_s": m_ / sn},
               "shards_full_file": full,
               "best_shards_full_file": max(full.values(), key=lambda x: x["reads_per_s"]),
               "kmahip_map_on_sample": {"wall_s": round(ws[0], 3), "reads_per_s": m_ / ws[0]},
               "res_identical_to_reference": same,
               "note": "the reference binary (oracle/_ref/kma -1t1) file to file on the first sample_reads reads of the e2e FASTQ: one thread, "
                       "-t nproc, and independent -t 1 processes over equal shards of the sample (wall = the slowest; every process "
                       "loads the index itself, so short shards are start-up bound); shards_full_file: the same arrangement over ALL reads "
                       "of the file, the figure vs_reference_best_shards is taken against (N independent processes do not produce the "
                       "single run's files -- ConClave sees one shard each --, they are the reference's speed limit, not a substitute)"}
        say(f"e2e: reference on {m_} reads: -t 1 {t1:.1f} s, -t {nproc} {tn:.1f} s, 16 shards {s16:.1f} s, {nproc} shards {sn:.1f} s; .res identical {same}")
        # the reference's DEFAULT mode (no -1t1: chain finder, reads mapping in pieces) on the same sample, both sides file to file
        try:
            t0 = time.perf_counter()
            r = subprocess.run([mapper, "-i", sfq, "-t_db", prefix, "-o", got + "_chain", "-chain"], stderr=subprocess.PIPE)
            tc = time.perf_counter() - t0
            if r.returncode:
                raise RuntimeError(r.stderr.decode().strip().splitlines()[-1] if r.stderr else "kmahip_map -chain failed")
            k_ = min(m_, 200_000)
            cfq = os.path.join(tmp, "e2e_chain_sample.fq")
            with open(sfq, "rb") as f, open(cfq, "wb") as g:
                g.write(f.read(rec * k_))
            t0 = time.perf_counter()
            subprocess.run([kma, "-i", cfq, "-o", os.path.join(tmp, "e2e_ref_chain"), "-t_db", prefix, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            trc = time.perf_counter() - t0
            subprocess.run([mapper, "-i", cfq, "-t_db", prefix, "-o", got + "_chain_s", "-chain"], check=True, stderr=subprocess.DEVNULL)
            ref["default_mode"] = {"what": "no -1t1 (save_kmers_chain): examples/kmahip_map -chain file to file on sample_reads reads; the reference -t 1 on the first reference_reads of them",
                                   "kmahip_map_chain": {"wall_s": round(tc, 3), "reads_per_s": m_ / tc},
                                   "reference_t1": {"reference_reads": k_, "wall_s": round(trc, 2), "reads_per_s": k_ / trc},
                                   "res_identical_to_reference": open(got + "_chain_s.res", "rb").read() == open(os.path.join(tmp, "e2e_ref_chain.res"), "rb").read()}
            say(f"e2e: default mode: kmahip_map -chain {tc:.2f} s on {m_} reads; reference {trc:.1f} s on {k_}; .res identical {ref['default_mode']['res_identical_to_reference']}")
        except Exception as e:  # noqa: BLE001  (extra figure only)
            ref["default_mode"] = {"error": str(e)}
        # paired end (`-ipe r1 r2 -apm p -1t1`, the shape of configs C3): both sides file to file
        try:
            from kma_amd import synth_dev
            n_pairs, k_ = max(1, pe_pairs), min(m_ // 2, 100_000, max(1, pe_pairs))
            r1, r2 = os.path.join(tmp, "e2e_r1.fq"), os.path.join(tmp, "e2e_r2.fq")
            t0 = time.perf_counter()
            with open(r1, "wb") as f1, open(r2, "wb") as f2:
                for c0, both in synth_dev.iter_pair_codes(seqs, n_pairs, seed=500, device=device, chunk=1 << 20):
                    hb = both.cpu().numpy()
                    for f_, mm in ((f1, hb[0::2]), (f2, hb[1::2])):
                        write_fastq_fixed(os.path.join(tmp, "part.fq"), mm)
                        with open(os.path.join(tmp, "part.fq"), "rb") as g:
                            shutil.copyfileobj(g, f_, 1 << 24)
            os.unlink(os.path.join(tmp, "part.fq"))
            say(f"e2e: two FASTQ files of {n_pairs} pairs, {2 * os.path.getsize(r1) / 1e9:.2f} GB, written in {time.perf_counter() - t0:.1f} s")
            walls = []
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([mapper, "-ipe", r1, r2, "-t_db", prefix, "-o", got + "_pe", "-1t1", "-apm", "p"], stderr=subprocess.PIPE)
                walls.append(time.perf_counter() - t0)
                if r.returncode:
                    raise RuntimeError(r.stderr.decode().strip().splitlines()[-1] if r.stderr else "kmahip_map -ipe failed")
            rec2 = os.path.getsize(r1) // n_pairs
            subs = []
            for p_ in (r1, r2):
                with open(p_, "rb") as f, open(p_ + ".sample", "wb") as g:
                    g.write(f.read(rec2 * k_))
                subs.append(p_ + ".sample")
            t0 = time.perf_counter()
            subprocess.run([kma, "-ipe", subs[0], subs[1], "-o", os.path.join(tmp, "e2e_ref_pe"), "-t_db", prefix, "-1t1", "-apm", "p", "-t", "1"], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            trp = time.perf_counter() - t0
            subprocess.run([mapper, "-ipe", subs[0], subs[1], "-t_db", prefix, "-o", got + "_pe_s", "-1t1", "-apm", "p"], check=True, stderr=subprocess.DEVNULL)
            ref["paired_end"] = {"what": "examples/kmahip_map -ipe r1 r2 -1t1 (kmahip_run_pe: pairing penalty as -apm p) file to file on `pairs` pairs of 2 x 150 nt; "
                                         "the reference -ipe r1 r2 -apm p -1t1 -t 1 on the first reference_pairs of them",
                                 "kmahip_map_ipe": {"pairs": n_pairs, "wall_s": round(min(walls), 3), "reads_per_s": 2 * n_pairs / min(walls)},
                                 "reference_t1": {"reference_pairs": k_, "wall_s": round(trp, 2), "reads_per_s": 2 * k_ / trp},
                                 "res_identical_to_reference": open(got + "_pe_s.res", "rb").read() == open(os.path.join(tmp, "e2e_ref_pe.res"), "rb").read()}
            say(f"e2e: paired end: kmahip_map -ipe {min(walls):.2f} s on {n_pairs} pairs; reference {trp:.1f} s on {k_}; .res identical {ref['paired_end']['res_identical_to_reference']}")
            for p_ in (r1, r2, *subs):
                os.unlink(p_)
        except Exception as e:  # noqa: BLE001  (extra figure only)
            ref["paired_end"] = {"error": str(e)}
        # long reads in the default mode with -bcNano (no -Mt1: how ONT reads are commonly run), both sides file to file
        try:
            rng_ = np.random.default_rng(4)
            genome = rng_.integers(0, 4, 2_000_000, dtype=np.uint8)
            gp = os.path.join(tmp, "g2mb")
            synth.write_fasta(gp + ".fsa", ["genome2Mb"], [genome])
            subprocess.run([kma, "index", "-i", gp + ".fsa", "-o", gp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            n_long = 2000
            lfq = os.path.join(tmp, "ont.fq")
            synth.write_fastq(lfq, synth.make_long_reads(genome, n_long, read_len=10000, seed=8), prefix="r", qual=b"5")
            t0 = time.perf_counter()
            subprocess.run([kma, "-i", lfq, "-o", os.path.join(tmp, "e2e_ref_long"), "-t_db", gp, "-bcNano", "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            trl = time.perf_counter() - t0
            t0 = time.perf_counter()
            subprocess.run([mapper, "-i", lfq, "-t_db", gp, "-o", got + "_long", "-chain", "-bcNano"], check=True, stderr=subprocess.DEVNULL)
            tl = time.perf_counter() - t0
            ref["long_default_mode"] = {"what": "2 000 ONT-like reads of 10 kb (4/3/3 % sub/del/ins) against one 2 Mb genome, default mode with -bcNano: kmahip_map -chain -bcNano "
                                                "and the reference -bcNano -t 1, file to file",
                                        "kmahip_map": {"wall_s": round(tl, 3), "reads_per_s": n_long / tl}, "reference_t1": {"wall_s": round(trl, 2), "reads_per_s": n_long / trl},
                                        "res_identical_to_reference": open(got + "_long.res", "rb").read() == open(os.path.join(tmp, "e2e_ref_long.res"), "rb").read()}
            say(f"e2e: long reads, default mode -bcNano: kmahip_map {tl:.2f} s, reference {trl:.1f} s; .res identical {ref['long_default_mode']['res_identical_to_reference']}")
        except Exception as e:  # noqa: BLE001  (extra figure only)
            ref["long_default_mode"] = {"error": str(e)}
        out["vs_reference_t1"] = out["plain"]["reads_per_s"] / ref["t1"]["reads_per_s"]
        # the N-process best case is taken on the WHOLE file (the 1 M-read sample's shards are start-up bound and understate it)
        out["vs_reference_best_shards"] = out["plain"]["reads_per_s"] / max(ref["best_shards_full_file"]["reads_per_s"], ref["t_nproc"]["reads_per_s"])
        out["vs_reference_best_shards_on_sample"] = out["plain"]["reads_per_s"] / max(ref["shards_16"]["reads_per_s"], ref["shards_nproc"]["reads_per_s"], ref["t_nproc"]["reads_per_s"])
    for f_ in (fq, gz):
        os.unlink(f_)
    return out, ref


def pe_step_leg(db, prefix, seqs, n_pairs, dev, steps, warmup, tmp, cpu_pairs):
    """BASELINE config C3 beside the headline step (extra key, never `value`): n_pairs pairs of 2 x 150 nt resident in HBM, one step =
    stage 2 with pairing (`-apm p`: scan_prefilter_kernel + scan_se_kernel in all-candidates mode + pair_penalty_kernel,
    save_kmers_penaltyPair savekmers.c:3572-3777) + stage 3a (alnFragsPenaltyPE alnfrags.c:1596-1972) + the per-pair reduction into the
    ConClave vectors. Same contract as the headline: K timed steps between synchronisations, kernel times by HIP events, algorithmic
    bytes from a counter-enabled launch, the reference's own stage timers on a bounded sample of the same pairs."""
    import re
    import torch
    from kma_amd import synth_dev
    n = 2 * n_pairs
    keep = min(n_pairs, cpu_pairs)
    rd = synth_dev.make_packed_pairs(seqs, n_pairs, seed=4242, device=dev, keep_codes=keep)
    i32 = lambda m: torch.empty(m, dtype=torch.int32, device=dev)
    mate, rc, rc_flag, flag, n_hits, best, oflag, out_rc = (i32(n) for _ in range(8))
    R_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    tcap = 4 * n
    T, h_t, h_sc, h_s, h_e = (i32(tcap) for _ in range(5))
    kind = i32(n_pairs)
    D = int(db.info.DB_size)
    aln = torch.zeros(D, dtype=torch.int64, device=dev)
    uniq = torch.zeros(D, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        aln.zero_(); uniq.zero_()
        db.scan_pe_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], mate, rc, rc_flag, flag, R_off, T, stream=stream)
        db.align_pe_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], 150, mate, rc, rc_flag, flag, R_off, T,
                        n_hits, best, oflag, h_t, h_sc, h_s, h_e, aln, uniq, out_rc, kind, stream=stream)
    for _ in range(max(1, warmup)):
        step()
    db.status(stream)
    db.set_timing(True)
    for k_ in range(4):
        db.get_timing(k_)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = {name: db.get_timing(i) for i, name in ((2, "prefilter"), (0, "scan"), (3, "seed"), (1, "align"))}
    db.set_timing(False)
    db.status(stream)
    db.set_stats(True)
    db.scan_pe_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], mate, rc, rc_flag, flag, R_off, T, stream=stream)
    st = db.get_stats(stream)
    db.set_stats(False)
    step()
    torch.cuda.synchronize()
    total_T = int(R_off[-1].item())
    kinds = torch.bincount(kind, minlength=5).tolist()
    W = (150 + 31) // 32
    scan_s = tm["scan"][0] / 1e3 / max(1, tm["scan"][1])
    # SURVEY 8(d) bytes of the reference layout, as for the headline kernel: 12 B per resolved k-mer start, 2 B per list element,
    # the packed read in, the record fields and 8 B per candidate (template + score) out
    scan_bytes = 12 * (st.probes - st.prefilter_probes) + 2 * st.value_elems + st.active_strands * (8 * W + 4 + 8 + 8) + n * (4 + 4 + 8) + 8 * total_T
    out = {"pairs": n_pairs, "reads": n, "steps": steps, "ms_per_step": dt / steps * 1e3, "pairs_per_s": n_pairs * steps / dt, "reads_per_s": n * steps / dt,
           "unit": "reads/s", "dtype": "i32",
           "workload": f"{n_pairs} x 2 x 150 nt pairs (insert 250-450, 0.5 % substitutions) vs the same {len(seqs)}-gene DB, -ipe -apm p -1t1; one step = stage 2 with "
                       "the pairing penalty + stage 3a for couples and singly filed records, on pairs resident in HBM",
           "kernel_ms": {k_: (v[0] / max(1, v[1])) for k_, v in tm.items()},
           "record_tasks": total_T, "pair_kinds_0none_1proper_2unmated_3first_4second": kinds,
           "roofline": {"bound": "hbm", "kernel": "scan_se_kernel<., 1, .> (all-candidates mode)", "kernel_ms": scan_s * 1e3,
                        "achieved": scan_bytes / scan_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": scan_bytes / scan_s / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": scan_bytes, "traffic": None,
                        "probes": int(st.probes - st.prefilter_probes), "hash_probes": int(st.hash_probes - st.prefilter_probes)}}
    ref = os.path.join(ROOT, "oracle", "_ref", "kma")
    if keep and os.path.exists(ref):
        m1, m2 = rd["codes"]
        r1, r2 = os.path.join(tmp, "pe_s1.fq"), os.path.join(tmp, "pe_s2.fq")
        write_fastq_fixed(r1, m1); write_fastq_fixed(r2, m2)
        t0 = time.time()
        r = subprocess.run([ref, "-ipe", r1, r2, "-o", os.path.join(tmp, "pe_cpu"), "-t_db", prefix, "-1t1", "-apm", "p", "-t", "1", "-status", "-nc", "-na", "-nf"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True)
        wall = time.time() - t0
        err = r.stderr.decode()
        m2_ = re.search(r"ankering query:\s*([0-9.]+) s", err)
        m3_ = re.search(r"KMA mapping time\s*([0-9.]+) s", err)
        if m2_ and m3_:
            s2, s3 = float(m2_.group(1)), float(m3_.group(1))
            out["cpu_baseline"] = dict(value=2 * keep / (s2 + s3), unit="reads/s", cores=1, kind="reference",
                                       sample=f"{keep} of the step's pairs; reference kma -ipe -apm p -1t1 -t 1 -status: stage 2 (ankering) {s2:.2f} s + stage 3a "
                                              f"(mapping) {s3:.2f} s of CPU time (whole pipeline {wall:.1f} s wall)", stage2_s=s2, stage3a_s=s3)
        for f_ in (r1, r2):
            os.unlink(f_)
    del rd
    torch.cuda.empty_cache()
    return out


def parity_sample(prefix, codes, got_scan, got_hits):
    """Checker leg: the first reads of the step vs the CPU oracle (bit-exact), stages 2 and 3a."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from kma_amd import formats
    batch = formats.pack_fixed(codes)
    odb = oracle.OracleDB(prefix)
    e = odb.scan_se(batch)
    ok = all(np.array_equal(a, b) for a, b in zip(e, got_scan))
    o = odb.align_se(batch, *e)
    ok = ok and np.array_equal(o["n_hits"], got_hits["n_hits"]) and np.array_equal(o["best_score"], got_hits["best_score"])
    T_off = e[2]
    for i in np.nonzero(o["n_hits"] > 0)[0]:
        a, c = int(T_off[i]), int(o["n_hits"][i])
        for key in ("tmpl", "start", "end", "score"):
            ok = ok and np.array_equal(o[key][a:a + c], got_hits[key][a:a + c])
    return bool(ok)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.share_gpu:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # n_gpus is what the process group reports, not what was asked for
    world = dist.get_world_size() if world > 1 else 1

    from kma_amd import binding, formats, synth, synth_dev
    from kma_amd.dist import allreduce_scores
    tmp = tempfile.mkdtemp(prefix=f"kmabench{rank}_")
    try:
        names, seqs = synth.make_gene_db(a.families, 5, 600, 1500, 0.04, seed=12345)
        prefix = os.path.join(tmp, "db5k")
        formats.write_index(prefix, names, seqs)
        db = binding.KmaHipDB(prefix, device=local)
        n = a.reads
        if a.scaling == "strong":
            from kma_amd.dist import shard_bounds
            lo, hi = shard_bounds(a.reads, rank, world)
            n = hi - lo
        keep = min(n, a.cpu_sample) if rank == 0 else 0
        rd = synth_dev.make_packed_reads(seqs, n, seed=1000 + rank, device=dev, keep_codes=keep,
                                         random_frac=0.02 if a.hard else 0.0, junk_frac=0.02 if a.hard else 0.0)
        rc_flag = torch.empty(n, dtype=torch.int32, device=dev)
        flag = torch.empty(n, dtype=torch.int32, device=dev)
        T_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        tcap = 8 * n
        T = torch.empty(tcap, dtype=torch.int32, device=dev)
        n_hits = torch.empty(n, dtype=torch.int32, device=dev)
        best = torch.empty(n, dtype=torch.int32, device=dev)
        oflag = torch.empty(n, dtype=torch.int32, device=dev)
        h_t, h_sc, h_s, h_e = (torch.empty(tcap, dtype=torch.int32, device=dev) for _ in range(4))
        D = int(db.info.DB_size)
        aln = torch.zeros(D, dtype=torch.int64, device=dev)
        uniq = torch.zeros(D, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def step(m=n):
            # m < n: the first m reads of this rank's batch (CSR prefix views) -- the strong-scaling leg
            aln.zero_(); uniq.zero_()
            db.scan_se_dev(rd["seq"], rd["seq_off"][:m + 1], rd["length"][:m], rd["N"], rd["N_off"][:m + 1], rc_flag[:m], flag[:m],
                           T_off[:m + 1], T, stream=stream)
            db.align_se_dev(rd["seq"], rd["seq_off"][:m + 1], rd["length"][:m], rd["N"], rd["N_off"][:m + 1], 150, rc_flag[:m], flag[:m],
                            T_off[:m + 1], T, n_hits[:m], best[:m], oflag[:m], h_t, h_sc, h_s, h_e, aln, uniq, stream=stream)
            # the path's only exchange: SUM of the two ConClave score vectors over the read shards (no-op at N=1)
            if world > 1 and a.backend != "nccl":
                ca, cu = aln.cpu(), uniq.cpu()
                allreduce_scores(ca, cu)
                aln.copy_(ca); uniq.copy_(cu)
            else:
                allreduce_scores(aln, uniq)

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
        scan_ms, scan_n = db.get_timing(0)
        aln_ms, aln_n = db.get_timing(1)
        pre_ms, pre_n = db.get_timing(2)
        seed_ms, seed_n = db.get_timing(3)
        db.set_timing(False)
        db.status(stream)

        def max_over_ranks(x):
            if world == 1:
                return x
            t = torch.tensor([x], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        dt = max_over_ranks(dt)
        # strong scaling beside the weak figure: the same K steps on this rank's share of ONE batch of a.reads reads
        strong = None
        if a.scaling == "weak" and world > 1:
            from kma_amd.dist import shard_bounds
            lo, hi = shard_bounds(a.reads, rank, world)
            m = hi - lo
            step(m)
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step(m)
            fence()
            dts = max_over_ranks(time.perf_counter() - t0)
            strong = {"reads_total": a.reads, "reads_per_gpu": m, "ms_per_step": dts / a.steps * 1e3, "value": a.reads * a.steps / dts,
                      "unit": "reads/s", "note": "same timed loop with the fixed batch sharded over the ranks (max over ranks)"}

        # algorithmic work of one launch (separate, untimed, counter-enabled launches)
        db.set_stats(True)
        db.scan_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], rc_flag, flag, T_off, T, stream=stream)
        st = db.get_stats(stream)
        db.align_se_dev(rd["seq"], rd["seq_off"], rd["length"], rd["N"], rd["N_off"], 150, rc_flag, flag, T_off, T,
                        n_hits, best, oflag, h_t, h_sc, h_s, h_e, aln, uniq, stream=stream)
        ast = db.get_align_stats(stream)
        db.set_stats(False)
        cand = int((T_off[1:] > T_off[:-1]).sum().item())
        mapped = int((n_hits > 0).sum().item())
        ties = int((n_hits < 0).sum().item())
        total_T = int(T_off[-1].item())
        W = (150 + 31) // 32
        scan_s = scan_ms / 1e3 / max(1, scan_n)
        aln_s = aln_ms / 1e3 / max(1, aln_n)
        pre_s = pre_ms / 1e3 / max(1, pre_n)
        seed_s = seed_ms / 1e3 / max(1, seed_n)
        # SURVEY §8(d) algorithmic bytes.  scan: 12 B per probe (4 B directory + 4 B key + 4 B value index in the
        # reference layout) + 2 B per value-list element + packed read in + S2 fields out.
        # Stage 2 runs as two kernels: scan_prefilter_kernel (every k-th k-mer of both strands; reads in, active list
        # out) and scan_se_kernel (all k-mer starts of the surviving strands; the rest of the bytes).
        pre_bytes = 12 * st.prefilter_probes + n * (8 * W + 4 + 8) + 8 * st.active_strands
        scan_bytes = (12 * (st.probes - st.prefilter_probes) + 2 * st.value_elems + st.active_strands * (8 * W + 4 + 8 + 8)
                      + n * (4 + 4 + 8) + 4 * total_T)
        # align: 12 B per position-index lookup (4 B index + 8 B template word), 2 bits per MEM base on both
        # sequences, packed read in per task, 24 B out per task; DP cells move no HBM bytes (no E matrix)
        # Stage 3a runs as seed_tasks_kernel (the MEM search: lookups + MEM bases + packed read in, 4 + 32 B handed over per
        # task) and align_tasks_kernel (chain, stitch, DP: hand-over + packed read in, 24 B out per task)
        seed_bytes = 12 * ast.lookups + ast.mem_bases // 2 + ast.tasks * (8 * W + 36)
        aln_bytes = ast.tasks * (8 * W + 36 + 24) + n * 12
        if aln_s >= scan_s:
            dom = dict(kernel="align_tasks_kernel", kernel_ms=aln_s * 1e3, achieved=aln_bytes / aln_s / 1e9,
                       algorithmic_bytes_per_launch=aln_bytes)
        else:
            dom = dict(kernel="scan_se_kernel", kernel_ms=scan_s * 1e3, achieved=scan_bytes / scan_s / 1e9,
                       algorithmic_bytes_per_launch=scan_bytes)

        # HBM-side traffic of the dominant kernel: PMC counters cannot be read from inside this process; the committed rocprofv3
        # --pmc summary of this same command supplies it -- but only while the kernel sources are the ones it was taken on
        # (tools/collect_profiles.py stores their hash); a stale figure is dropped, not reported
        traffic, traffic_src = None, None
        for prof in ("r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", prof)))
                if dom["kernel"] in pmc and n == 10_000_000 and a.families == 1000 and pmc.get("_kernel_src_sha256") == kernel_src_sha():
                    traffic = pmc[dom["kernel"]]["corrected_bytes"]
                    traffic_src = f"profiles/{prof}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes, kernel sources {pmc['_kernel_src_sha256'][:12]}"
                    break
            except (OSError, ValueError, KeyError):
                pass

        out = {
            "metric": "mapped reads/sec (whole node), 10M×150bp vs 5k-gene DB, 1/2/4/8 GPU",
            "value": (world * n if a.scaling == "weak" else a.reads) * a.steps / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": f"{n} x 150 bp SE reads per GPU" + (f" ({a.reads} in total, sharded)" if a.scaling == "strong" else "") + f" vs {5 * a.families}-gene DB (k=16), -1t1; one step = "
                            "stage 2 (k-mer probe + candidate-template scoring) + stage 3a (MEM seeding, chaining, "
                            "NW extension, per-read hit selection, ConClave score vectors) on reads resident in HBM"
                            + ("; HARD MIX (not the BASELINE workload): 2 % unmappable reads, 2 % with 60-120 foreign end bases" if a.hard else ""),
                "reads_per_gpu": n, "genes": 5 * a.families, "db_kmers": int(db.info.n_kmers),
                "probe_table_MB": round(db.info.hash_bytes / 1e6, 1), "db_total_MB": round(db.info.total_bytes / 1e6, 1),
                "stage2_candidate_fraction": cand / n, "mapped_fraction": mapped / n, "strand_tie_reads": ties,
                "align_tasks": total_T, "parallelism": f"read-shard x{world}",
            },
            "roofline": {
                "bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["achieved"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "achieved_kind": "effective: SURVEY 8(d) algorithmic bytes of the REFERENCE layout (12 B per resolved k-mer start, ...) / "
                                 "kernel time measured with HIP events in this run; the kernel resolves most k-mer starts by walking the "
                                 "template store instead of probing (see scan.hash_probes), so this is not a DRAM bandwidth reading",
                "kernel": dom["kernel"], "kernel_ms": dom["kernel_ms"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "prefilter": {"kernel_ms": pre_s * 1e3, "probes": int(st.prefilter_probes), "GB/s": pre_bytes / pre_s / 1e9,
                              "active_strands": int(st.active_strands)},
                "scan": {"kernel_ms": scan_s * 1e3, "probes": int(st.probes - st.prefilter_probes),
                         "hash_probes": int(st.hash_probes - st.prefilter_probes),
                         "GB/s": scan_bytes / scan_s / 1e9, "probes_per_s": (st.probes - st.prefilter_probes) / scan_s},
                "seed": {"kernel_ms": seed_s * 1e3, "lookups": int(ast.lookups), "GB/s": seed_bytes / seed_s / 1e9 if seed_s else None},
                "align": {"kernel_ms": aln_s * 1e3, "lookups": int(ast.lookups), "dp_cells": int(ast.dp_cells),
                          "GCUPS": ast.dp_cells / aln_s / 1e9, "GB/s": aln_bytes / aln_s / 1e9,
                          "tasks_per_s": ast.tasks / aln_s},
            },
        }
        # what carried the step's one exchange, as the transport reports it (a SCALE line can be checked for "RCCL saw N ranks")
        comm = {"backend": "none", "nranks": 1}
        if world > 1:
            comm = {"backend": dist.get_backend(), "nranks": dist.get_world_size(), "rank0_device": str(dev),
                    "exchange_per_step": f"SUM all-reduce of alignment_scores and uniq_alignment_scores: 2 x u64[{D}]"}
            if a.backend == "nccl":
                try:
                    comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
                except Exception:  # noqa: BLE001
                    pass
                # every rank adds 1: the sum IS the number of ranks the collective reached
                one = torch.ones(1, dtype=torch.int64, device=dev)
                dist.all_reduce(one)
                comm["ranks_counted_by_allreduce"] = int(one.item())
        out["config"]["comm"] = comm
        if rank == 0 and world == 1 and not a.no_cpu and keep:
            codes = rd["codes"][:keep]
            k = min(a.parity_sample, keep)
            got = [x.cpu().numpy() for x in (rc_flag[:k], flag[:k], T_off[:k + 1])]
            nt = int(got[2][-1])
            got.append(T[:nt].cpu().numpy())
            hits = dict(n_hits=n_hits[:k].cpu().numpy(), best_score=best[:k].cpu().numpy(), tmpl=h_t[:nt].cpu().numpy(),
                        score=h_sc[:nt].cpu().numpy(), start=h_s[:nt].cpu().numpy(), end=h_e[:nt].cpu().numpy())
            out["config"]["parity_first_reads_vs_oracle"] = parity_sample(prefix, codes[:k], got, hits)
            out["config"]["parity_reads_checked"] = k
            # (the device legs first, the reference's CPU runs after them: behind those, copies to the device and the long-read leg --
            # hundreds of launches, host syncs between them -- came out up to twice as slow in some runs of this file and not in others)
            if a.pe_pairs > 0 and not a.hard:
                try:
                    out["paired_end_step"] = pe_step_leg(db, prefix, seqs, a.pe_pairs, dev, a.steps, a.warmup, tmp, min(a.cpu_sample // 2, 500_000))
                except Exception as e:  # noqa: BLE001  (extra leg only)
                    out["paired_end_step"] = {"error": str(e)}
            # beyond the benchmarked step (informational, never `value`): the same sample through kmahip_run_se -- host
            # buffers in, `.res` statistics + consensus out: upload, stages 2 + 3a, ConClave, traceback, pile-up, consensus
            try:
                # the step's own batch, copied back to host buffers: all n reads (the metric's 10 M), as a caller holding packed reads has them
                pb = formats.ReadBatch(rd["seq"].cpu().numpy().view(np.uint64), rd["seq_off"].cpu().numpy(), rd["length"].cpu().numpy(),
                                       rd["N"].cpu().numpy()[:0], rd["N_off"].cpu().numpy())
                db.run_se(pb, per_read=False)          # first call: scratch allocation of stage 3c (10 GB for the trace lanes)
                t0 = time.perf_counter()
                o = db.run_se(pb, per_read=False)
                dt_p = time.perf_counter() - t0
                out["whole_pipeline"] = {"reads": int(pb.n), "reads_per_s": pb.n / (sum(o["ms"]) / 1e3), "call_wall_ms": dt_p * 1e3,
                                         "stage_ms": {k: round(v, 2) for k, v in zip(("upload", "stage2+3a", "conclave+stats", "traceback",
                                                                                      "pileup+consensus", "copies"), o["ms"])},
                                         "res_rows": sum(1 for r in o["rows"] if r.significant),
                                         "note": "kmahip_run_se on the reads of the timed step (packed reads in HOST memory -> per-template results: PCIe "
                                                 "upload, stages 2 + 3a, ConClave + .res statistics, traceback, pile-up, consensus); reads_per_s = reads / "
                                                 "sum(stage_ms); call_wall_ms adds the Python-side result buffers"}
            except Exception as e:  # noqa: BLE001  (informational leg only)
                out["whole_pipeline"] = {"error": str(e)}
            if a.c4_reads > 0 and not a.hard:
                try:
                    out["c4"] = c4_leg(tmp, a.c4_reads, local, a.c4_parity)
                except Exception as e:  # noqa: BLE001  (extra leg only)
                    out["c4"] = {"error": str(e)}
            if not a.hard:
                try:
                    out["default_mode_stage2"] = chain_stage2_leg(db, pb, tmp, local)
                except Exception as e:  # noqa: BLE001  (extra leg only)
                    out["default_mode_stage2"] = {"error": str(e)}
            out["cpu_baseline"] = cpu_baseline(prefix, codes, tmp)
            if a.e2e_reads > 0 and not a.hard:
                try:
                    db.close()            # the C host program opens the index itself; give it the card's memory back first
                    torch.cuda.empty_cache()
                    e2e, ref = e2e_leg(tmp, prefix, seqs, a.e2e_reads, a.e2e_sample, pe_pairs=a.pe_pairs if a.pe_pairs > 0 else 1_000_000, device=dev)
                    out["e2e"] = e2e
                    if ref is not None and out.get("cpu_baseline"):
                        out["cpu_baseline"].update(cpu_model=ref["cpu_model"], nproc=ref["nproc"], file_to_file=ref)
                except Exception as e:  # noqa: BLE001  (extra leg only)
                    out["e2e"] = {"error": str(e)}
        elif rank == 0:
            out["cpu_baseline"] = None
        if strong is not None:
            out["strong_scaling"] = strong
        if rank == 0:
            print(json.dumps(out), flush=True)
        db.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
