#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the compiled reference.

Run in the build container (needs oracle/_ref/kma, built by `make -C oracle ref`
from the sources under /root/reference).  Inputs are synthetic and seeded; the
outputs are the reference's own stream taps:
    -s1            stage-1 records (2-bit packed reads)
    -s2            stage-2 records (candidate templates, k-mer score, strand)
    -a             frag_raw text (per-read alignment picks after stage 3a)
    .res/.frag.gz  final outputs
Only data is stored: FASTA/FASTQ inputs, the index files written by
`kma index`, and the tapped outputs.
"""
import gzip
import lzma
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from kma_amd import synth  # noqa: E402

KMA = os.path.join(ROOT, "oracle", "_ref", "kma")


def tricky_db(seed):
    rng = np.random.default_rng(seed)
    names, seqs = synth.make_gene_db(12, 5, 400, 1200, 0.04, seed)
    # exact duplicate of an existing template (equal value sets everywhere)
    names.append("dup_of_fam00003_v0"); seqs.append(seqs[15].copy())
    # internal tandem repeat: duplicated k-mers inside one template
    unit = rng.integers(0, 4, 90, dtype=np.uint8)
    rep = np.concatenate([rng.integers(0, 4, 200, dtype=np.uint8), unit, unit, unit,
                          rng.integers(0, 4, 150, dtype=np.uint8)])
    names.append("tandem_repeat"); seqs.append(rep)
    # low complexity: poly-A stretch (k-mer 0 is never position-indexed)
    pa = np.concatenate([rng.integers(0, 4, 180, dtype=np.uint8), np.zeros(60, np.uint8),
                         rng.integers(0, 4, 180, dtype=np.uint8)])
    names.append("polyA_island"); seqs.append(pa)
    # short template (shorter than a read)
    names.append("short_template"); seqs.append(rng.integers(0, 4, 120, dtype=np.uint8))
    # chimera of two families (a read can bridge value sets)
    names.append("chimera_f1_f7"); seqs.append(np.concatenate([seqs[5][:300], seqs[35][100:420]]))
    # the same gene indexed on both strands, and an inverted repeat: reads from these tie between
    # the forward and reverse strand in stage 2 (rc_flag < 0 -> anker_rc_comp path in stage 3a)
    names.append("rc_of_fam00002_v0"); seqs.append(synth.revcomp_codes(seqs[10]).copy())
    x = rng.integers(0, 4, 320, dtype=np.uint8)
    names.append("inverted_repeat"); seqs.append(np.concatenate([x, rng.integers(0, 4, 25, dtype=np.uint8), synth.revcomp_codes(x)]))
    return names, seqs


def tricky_reads(seqs, n, seed):
    rng = np.random.default_rng(seed)
    out = []
    lens = np.array([len(s) for s in seqs])
    for i in range(n):
        kind = rng.random()
        g = int(rng.integers(0, len(seqs)))
        s = seqs[g]
        if kind < 0.05:     # random, unmappable
            r = rng.integers(0, 4, int(rng.integers(30, 200)), dtype=np.uint8)
        else:
            L = int(rng.choice([150, 150, 150, 100, 75, 250, 40, 17, 16, 15]))
            if kind < 0.20:  # overhang past template ends with random flanks
                left = rng.integers(0, 4, int(rng.integers(0, 40)), dtype=np.uint8)
                right = rng.integers(0, 4, int(rng.integers(0, 40)), dtype=np.uint8)
                if rng.random() < 0.5:
                    core = s[:max(16, L - len(left))]
                    r = np.concatenate([left, core])
                else:
                    core = s[-max(16, L - len(right)):]
                    r = np.concatenate([core, right])
            else:
                L = min(L, len(s))
                st = int(rng.integers(0, len(s) - L + 1))
                r = s[st:st + L].copy()
            # substitutions
            rate = float(rng.choice([0.0, 0.005, 0.01, 0.03, 0.06]))
            m = rng.random(len(r)) < rate
            r = r.copy()
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            # indels
            if rng.random() < 0.25 and len(r) > 40:
                for _ in range(int(rng.integers(1, 3))):
                    p = int(rng.integers(10, len(r) - 10))
                    ln = int(rng.integers(1, 5))
                    if rng.random() < 0.5:
                        r = np.delete(r, slice(p, p + ln))
                    else:
                        r = np.insert(r, p, rng.integers(0, 4, ln, dtype=np.uint8))
            # Ns
            if rng.random() < 0.15:
                m = rng.random(len(r)) < float(rng.choice([0.01, 0.05]))
                r[m] = 4
            if rng.random() < 0.5:
                r = synth.revcomp_codes(r)
        out.append(np.ascontiguousarray(r.astype(np.uint8)))
    return out


def run(cmd, stdout=None):
    subprocess.run(cmd, check=True, stdout=stdout, stderr=subprocess.DEVNULL)


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def xz(src, dst):
    with open(src, "rb") as f, lzma.open(dst, "wb", preset=9) as g:
        shutil.copyfileobj(f, g)


def make_se(outdir, seed=7):
    os.makedirs(outdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        names, seqs = tricky_db(seed)
        fa = os.path.join(tmp, "db.fsa")
        synth.write_fasta(fa, names, seqs)
        reads = tricky_reads(seqs, 1500, seed + 1)
        fq = os.path.join(tmp, "reads.fq")
        synth.write_fastq(fq, reads)
        db = os.path.join(tmp, "db")
        run([KMA, "index", "-i", fa, "-o", db])
        base = [KMA, "-i", fq, "-o", os.path.join(tmp, "out"), "-t_db", db, "-1t1", "-t", "1"]
        with open(os.path.join(tmp, "s1.bin"), "wb") as f:
            run(base + ["-s1"], stdout=f)
        with open(os.path.join(tmp, "s2.bin"), "wb") as f:
            run(base + ["-s2"], stdout=f)
        with open(os.path.join(tmp, "s2_ex.bin"), "wb") as f:
            run(base + ["-s2", "-ex_mode"], stdout=f)
        run(base + ["-a"])
        gz(fa, os.path.join(outdir, "db.fsa.gz"))
        gz(fq, os.path.join(outdir, "reads.fq.gz"))
        xz(db + ".comp.b", os.path.join(outdir, "db.comp.b.xz"))
        for ext in (".length.b", ".seq.b", ".name"):
            shutil.copy(db + ext, os.path.join(outdir, "db" + ext))
        for b in ("s1.bin", "s2.bin", "s2_ex.bin"):
            gz(os.path.join(tmp, b), os.path.join(outdir, b + ".gz"))
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(outdir, "out.res"))
        shutil.copy(os.path.join(tmp, "out.frag.gz"), os.path.join(outdir, "out.frag.gz"))
        shutil.copy(os.path.join(tmp, "out.frag_raw.gz"), os.path.join(outdir, "out.frag_raw.gz"))


def long_reads(seqs, n, seed):
    """Long reads that force banded NW: divergent internal blocks (no shared
    16-mer for 70-160 bases), junk flanks > 128 bases, indel-rich stretches."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(350, min(1400, len(s))))
        st = int(rng.integers(0, len(s) - L + 1))
        r = s[st:st + L].copy()
        kind = rng.random()
        if kind < 0.45:      # divergent block(s): every 7th-9th base substituted
            for _ in range(int(rng.integers(1, 3))):
                b0 = int(rng.integers(40, L - 200)); bl = int(rng.integers(70, 160))
                step = int(rng.integers(5, 10))
                idx = np.arange(b0, min(L - 30, b0 + bl), step)
                r[idx] = (r[idx] + rng.integers(1, 4, len(idx), dtype=np.uint8)) & 3
                if rng.random() < 0.5:   # plus a length change inside the block
                    p = b0 + bl // 2
                    if rng.random() < 0.5:
                        r = np.delete(r, slice(p, p + int(rng.integers(1, 12))))
                    else:
                        r = np.insert(r, p, rng.integers(0, 4, int(rng.integers(1, 12)), dtype=np.uint8))
                    L = len(r)
        elif kind < 0.75:    # junk flanks
            a = rng.integers(0, 4, int(rng.integers(100, 260)), dtype=np.uint8)
            b = rng.integers(0, 4, int(rng.integers(0, 260)), dtype=np.uint8)
            r = np.concatenate([a, r, b]) if rng.random() < 0.5 else np.concatenate([b, r, a])
        else:                # ONT-like
            m = rng.random(len(r)) < 0.04
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            keep = rng.random(len(r)) >= 0.02
            r = r[keep]
            ip = np.nonzero(rng.random(len(r)) < 0.02)[0]
            r = np.insert(r, ip, rng.integers(0, 4, len(ip), dtype=np.uint8))
        if rng.random() < 0.1:
            m = rng.random(len(r)) < 0.004
            r[m] = 4
        if rng.random() < 0.5:
            r = synth.revcomp_codes(r)
        out.append(np.ascontiguousarray(r.astype(np.uint8)))
    return out


def make_long(outdir, seed=21):
    os.makedirs(outdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        names, seqs = synth.make_gene_db(6, 3, 1800, 2600, 0.03, seed)
        fa = os.path.join(tmp, "db.fsa")
        synth.write_fasta(fa, names, seqs)
        reads = long_reads(seqs, 260, seed + 1)
        fq = os.path.join(tmp, "reads.fq")
        synth.write_fastq(fq, reads)
        db = os.path.join(tmp, "db")
        run([KMA, "index", "-i", fa, "-o", db])
        base = [KMA, "-i", fq, "-o", os.path.join(tmp, "out"), "-t_db", db, "-1t1", "-t", "1"]
        with open(os.path.join(tmp, "s1.bin"), "wb") as f:
            run(base + ["-s1"], stdout=f)
        with open(os.path.join(tmp, "s2.bin"), "wb") as f:
            run(base + ["-s2"], stdout=f)
        run(base + ["-a"])
        gz(fa, os.path.join(outdir, "db.fsa.gz"))
        gz(fq, os.path.join(outdir, "reads.fq.gz"))
        xz(db + ".comp.b", os.path.join(outdir, "db.comp.b.xz"))
        for ext in (".length.b", ".seq.b", ".name"):
            shutil.copy(db + ext, os.path.join(outdir, "db" + ext))
        for b in ("s1.bin", "s2.bin"):
            gz(os.path.join(tmp, b), os.path.join(outdir, b + ".gz"))
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(outdir, "out.res"))
        shutil.copy(os.path.join(tmp, "out.frag_raw.gz"), os.path.join(outdir, "out.frag_raw.gz"))


def tricky_pairs(seqs, n, seed):
    """Read pairs: proper pairs (either orientation), mates on different templates, one mate random,
    short / N-containing mates, overlapping mates."""
    rng = np.random.default_rng(seed)
    m1, m2 = [], []
    for i in range(n):
        kind = rng.random()
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.choice([150, 150, 100, 75, 40]))
        ins = int(rng.integers(max(L, 60), 500))
        ins = min(ins, len(s))
        L = min(L, ins)
        st = int(rng.integers(0, len(s) - ins + 1))
        frag = s[st:st + ins]
        a = frag[:L].copy()
        b = synth.revcomp_codes(frag[-L:]).copy()
        if kind < 0.12:       # second mate from another template
            s2 = seqs[int(rng.integers(0, len(seqs)))]
            st2 = int(rng.integers(0, max(1, len(s2) - L)))
            b = s2[st2:st2 + L].copy()
            if rng.random() < 0.5:
                b = synth.revcomp_codes(b)
        elif kind < 0.20:     # one mate random
            if rng.random() < 0.5:
                a = rng.integers(0, 4, L, dtype=np.uint8)
            else:
                b = rng.integers(0, 4, L, dtype=np.uint8)
        elif kind < 0.25:     # same strand (improper orientation)
            b = frag[-L:].copy()
        for r in (a, b):
            rate = float(rng.choice([0.0, 0.005, 0.02, 0.05]))
            m = rng.random(len(r)) < rate
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            if rng.random() < 0.1:
                m = rng.random(len(r)) < 0.02
                r[m] = 4
        if rng.random() < 0.08:
            a = a[: int(rng.integers(10, 30))]
        if rng.random() < 0.5:
            a, b = b, a
        m1.append(np.ascontiguousarray(a)); m2.append(np.ascontiguousarray(b))
    return m1, m2


def make_pe(outdir, seed=33):
    os.makedirs(outdir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        names, seqs = tricky_db(seed)
        fa = os.path.join(tmp, "db.fsa")
        synth.write_fasta(fa, names, seqs)
        m1, m2 = tricky_pairs(seqs, 900, seed + 1)
        fq1, fq2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
        synth.write_fastq(fq1, m1, prefix="p")
        synth.write_fastq(fq2, m2, prefix="p")
        db = os.path.join(tmp, "db")
        run([KMA, "index", "-i", fa, "-o", db])
        base = [KMA, "-ipe", fq1, fq2, "-o", os.path.join(tmp, "out"), "-t_db", db, "-1t1", "-apm", "p", "-t", "1"]
        with open(os.path.join(tmp, "s1.bin"), "wb") as f:
            run(base + ["-s1"], stdout=f)
        with open(os.path.join(tmp, "s2.bin"), "wb") as f:
            run(base + ["-s2"], stdout=f)
        run(base + ["-a"])
        gz(fa, os.path.join(outdir, "db.fsa.gz"))
        gz(fq1, os.path.join(outdir, "r1.fq.gz")); gz(fq2, os.path.join(outdir, "r2.fq.gz"))
        xz(db + ".comp.b", os.path.join(outdir, "db.comp.b.xz"))
        for ext in (".length.b", ".seq.b", ".name"):
            shutil.copy(db + ext, os.path.join(outdir, "db" + ext))
        for b in ("s1.bin", "s2.bin"):
            gz(os.path.join(tmp, b), os.path.join(outdir, b + ".gz"))
        shutil.copy(os.path.join(tmp, "out.res"), os.path.join(outdir, "out.res"))
        shutil.copy(os.path.join(tmp, "out.frag_raw.gz"), os.path.join(outdir, "out.frag_raw.gz"))


if __name__ == "__main__":
    if not os.path.exists(KMA):
        sys.exit("oracle/_ref/kma missing: run `make -C oracle ref` first")
    make_se(os.path.join(HERE, "se"))
    make_long(os.path.join(HERE, "long"))
    make_pe(os.path.join(HERE, "pe"))
    print("golden fixtures written")
