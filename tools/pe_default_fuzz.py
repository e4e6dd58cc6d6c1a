#!/usr/bin/env python3
"""Parity beyond the test suite for the paired run in the reference's DEFAULT mode (no -1t1: couples through save_kmers_pair, records
that lost their mate through the chain finder -- kmahip_ws_set_pe_chain) and for the switches round 4 added to examples/kmahip_map:
seeded inputs with ragged read lengths, N's, indels, foreign mates, trimmed-away mates whose partner is made of pieces of several
genes, given as two mate files and interleaved, under random pairing modes, -mrc, -mf and scoring schemes, through the compiled
reference (-t 1) and through kmahip_map; `.res`, `.fsa`, `.aln` and the inflated `.frag.gz` must be the same bytes.

    gpurun -- 'python3 tools/pe_default_fuzz.py 12'          (needs oracle/_ref/kma)
"""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kma_amd import formats, synth  # noqa: E402

KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
MAP = os.path.join(ROOT, "examples", "kmahip_map")
LUT = np.frombuffer(b"ACGTN", dtype=np.uint8)


def case(tmp, seed):
    rng = np.random.default_rng(seed)
    names, seqs = synth.make_gene_db(int(rng.integers(10, 40)), int(rng.integers(2, 6)), 500, 1500, 0.04, seed=1000 + seed)
    if seed % 3 == 0:          # templates that are the reverse complement of others: strand ties
        for g in range(0, len(seqs), 7):
            names.append("rc_of_" + names[g]); seqs.append((3 - seqs[g])[::-1].copy())
    prefix = os.path.join(tmp, "db")
    formats.write_index(prefix, names, seqs)
    n_pairs = int(rng.integers(1500, 5000))
    m1, m2, _ = synth.make_pairs(seqs, n_pairs, seed=2000 + seed)
    r1, r2 = [r.copy() for r in m1], [r.copy() for r in m2]
    q1, q2 = [bytearray(b"I" * len(r)) for r in r1], [bytearray(b"I" * len(r)) for r in r2]
    for i in rng.choice(n_pairs, n_pairs // 15, replace=False):          # a foreign mate
        (r1 if rng.random() < 0.5 else r2)[i] = rng.integers(0, 4, 150, dtype=np.uint8)
    for i in rng.choice(n_pairs, n_pairs // 10, replace=False):          # ragged: a mate cut short
        r, q = (r1, q1) if rng.random() < 0.5 else (r2, q2)
        L = int(rng.integers(20, 150))
        r[i] = r[i][:L]; q[i] = q[i][:L]
    for i in rng.choice(n_pairs, n_pairs // 12, replace=False):          # N's
        r = r1 if rng.random() < 0.5 else r2
        for p in rng.integers(0, len(r[i]), int(rng.integers(1, 4))):
            r[i][int(p)] = 4
    for i in rng.choice(n_pairs, n_pairs // 12, replace=False):          # an insertion or a deletion
        r, q = (r1, q1) if rng.random() < 0.5 else (r2, q2)
        if len(r[i]) < 140:
            continue
        a = int(rng.integers(30, 110))
        r[i] = np.concatenate([r[i][:a], rng.integers(0, 4, 2, dtype=np.uint8), r[i][a:len(r[i]) - 2]]) if rng.random() < 0.5 else np.concatenate([r[i][:a], r[i][a + 2:], r[i][:2]])
    for x, i in enumerate(rng.choice(n_pairs, n_pairs // 8, replace=False)):          # a mate the quality trim removes: its partner is a single record
        first = rng.random() < 0.5
        q = q1 if first else q2
        q[i][10:] = b"#" * (len(q[i]) - 10)
        if x % 2 == 0:                                                                # ... made of two or three genes, pieces forward and reversed
            pieces = []
            for _ in range(int(rng.integers(2, 4))):
                g = seqs[int(rng.integers(0, len(seqs)))]
                a = int(rng.integers(0, len(g) - 70))
                p = g[a:a + int(rng.integers(40, 70))]
                pieces.append(p if rng.random() < 0.5 else (3 - p)[::-1])
            keep = r2 if first else r1
            keep[i] = np.concatenate(pieces).astype(np.uint8)
            (q2 if first else q1)[i] = bytearray(b"I" * len(keep[i]))
    paths = [os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq"), os.path.join(tmp, "ilv.fq")]
    with open(paths[0], "wb") as f1, open(paths[1], "wb") as f2, open(paths[2], "wb") as fi:
        for i in range(n_pairs):
            a = b"@p%d/1\n" % i + LUT[r1[i]].tobytes() + b"\n+\n" + bytes(q1[i]) + b"\n"
            b = b"@p%d/2\n" % i + LUT[r2[i]].tobytes() + b"\n+\n" + bytes(q2[i]) + b"\n"
            f1.write(a); f2.write(b); fi.write(a + b)
    return prefix, paths, rng


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    bad = 0
    only = os.environ.get("FUZZ_ONLY")          # (FUZZ_ONLY=<seed>: that case alone, its differing lines printed)
    for seed in ([int(only)] if only else range(n)):
        with tempfile.TemporaryDirectory() as tmp:
            prefix, (r1, r2, ilv), rng = case(tmp, seed)
            single = len(sys.argv) > 2 and sys.argv[2] == "se"          # (second argument "se": the interleaved file as single-end input -- -1t1 / default mode, -mem_mode, -lc, -and)
            inp = ["-i", ilv] if single else (["-int", ilv] if seed % 4 == 3 else ["-ipe", r1, r2])
            opts = []
            if not single:
                opts += [[], ["-apm", "p"], ["-apm", "u"], ["-pm", "p"], ["-fpm", "p"]][int(rng.integers(0, 5))]
            elif rng.random() < 0.3:
                opts += ["-and"]
            if rng.random() < 0.4:
                opts += ["-mrc", "0.6"]
            if rng.random() < 0.4:
                opts += ["-mf", str(int(rng.integers(5, 2000)))]
            if rng.random() < 0.3:
                opts += ["-cge"]
            elif rng.random() < 0.3:
                opts += ["-reward", "2", "-gapopen", "6", "-gapextend", "2", "-transition", "2", "-transversion", "5", "-per", "11"]
            if rng.random() < 0.3:
                opts += ["-1t1"]
            if rng.random() < 0.3:
                opts += ["-mem_mode"]
            if single and "-1t1" in opts and rng.random() < 0.4:
                opts += ["-lc"]
            args = inp + ["-t_db", prefix] + opts
            ref, got = os.path.join(tmp, "ref"), os.path.join(tmp, "got")
            subprocess.run([KMA] + args + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            env = dict(os.environ)
            if seed % 2:
                env["KMAHIP_MAP_BATCH"] = str(int(rng.integers(300, 3000)))
            r = subprocess.run([MAP] + args + ["-o", got], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            diff = []
            if r.returncode:
                diff.append("exit %d: %s" % (r.returncode, r.stderr.decode()[-300:]))
            else:
                for ext, opener in ((".res", open), (".fsa", open), (".aln", open), (".frag.gz", gzip.open)):
                    if opener(got + ext, "rb").read() != opener(ref + ext, "rb").read():
                        diff.append(ext)
            if only and diff:
                for ext, opener in ((".res", open), (".frag.gz", gzip.open)):
                    A, B = opener(ref + ext, "rb").read().splitlines(), opener(got + ext, "rb").read().splitlines()
                    sa, sb = set(A), set(B)
                    print(ext, len(A), len(B), "only in the reference:", [x[:160] for x in A if x not in sb][:6], "only ours:", [x[:160] for x in B if x not in sa][:6])
            rows = gzip.open(ref + ".frag.gz").read().count(b"\n")
            print(f"seed {seed}: {' '.join(args[args.index('-t_db') + 2:]) or '(default)'} {inp[0]}: {rows} fragment rows: {'SAME' if not diff else 'DIFFERENT ' + str(diff)}", flush=True)
            bad += bool(diff)
    print(f"{n - bad} of {n} runs identical to the reference")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
