"""Deterministic synthetic inputs for the KMA hot path.

Data generation only (numpy): gene-family template databases, 150 bp
single-end reads, paired reads and ONT-like long reads, written as FASTA/FASTQ
for the reference binary and returned as numpy arrays for our own packing.
Nothing here is on the measured path.
"""
from __future__ import annotations

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def make_gene_db(n_families=1000, variants=5, len_lo=600, len_hi=1500,
                 max_div=0.04, seed=12345):
    """Return (names, seqs) -- seqs are uint8 code arrays (0..3).

    Each family is a uniform random sequence; variant v carries
    v/(variants-1)*max_div substitutions relative to variant 0 (SURVEY §8d:
    0/1/2/3/4 % for the 5-variant DB)."""
    rng = np.random.default_rng(seed)
    names, seqs = [], []
    for f in range(n_families):
        L = int(rng.integers(len_lo, len_hi + 1))
        base = rng.integers(0, 4, size=L, dtype=np.uint8)
        for v in range(variants):
            s = base.copy()
            div = max_div * v / max(1, variants - 1)
            nsub = int(round(div * L))
            if nsub:
                pos = rng.choice(L, size=nsub, replace=False)
                s[pos] = (s[pos] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
            names.append(f"fam{f:05d}_v{v}")
            seqs.append(s)
    return names, seqs


def write_fasta(path, names, seqs, width=0):
    with open(path, "wb") as fh:
        for n, s in zip(names, seqs):
            fh.write(b">" + n.encode() + b"\n")
            fh.write(BASES[s].tobytes() + b"\n")


def revcomp_codes(s):
    return _COMP[s[::-1]]


def make_reads(seqs, n_reads, read_len=150, sub_rate=0.005, rc_frac=0.5,
               random_frac=0.0, n_rate=0.0, seed=1):
    """Sample reads (uint8 codes 0..4, 4 = N) from the templates.

    Returns (reads [n, read_len] uint8, origin template index (0-based, -1 for
    random), start, is_rc)."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(s) for s in seqs])
    ok = np.nonzero(lens >= read_len)[0]
    cat = np.concatenate(seqs)
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    g = ok[rng.integers(0, len(ok), size=n_reads)]
    st = (rng.random(n_reads) * (lens[g] - read_len + 1)).astype(np.int64)
    idx = (offs[g] + st)[:, None] + np.arange(read_len)[None, :]
    reads = cat[idx]
    if sub_rate > 0:
        m = rng.random(reads.shape) < sub_rate
        reads[m] = (reads[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
    is_rc = rng.random(n_reads) < rc_frac
    reads[is_rc] = _COMP[reads[is_rc][:, ::-1]]
    if random_frac > 0:
        r = rng.random(n_reads) < random_frac
        reads[r] = rng.integers(0, 4, size=(int(r.sum()), read_len), dtype=np.uint8)
        g = np.where(r, -1, g)
    if n_rate > 0:
        m = rng.random(reads.shape) < n_rate
        reads[m] = 4
    return reads, g, st, is_rc


def write_fastq(path, reads, prefix="r", qual=b"I", lens=None):
    """reads: [n, L] uint8 codes (0..4) or list of arrays."""
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as fh:
        for i, r in enumerate(reads):
            if lens is not None:
                r = r[: lens[i]]
            fh.write(b"@" + prefix.encode() + str(i).encode() + b"\n")
            fh.write(lut[r].tobytes() + b"\n+\n" + qual * len(r) + b"\n")


def make_pairs(seqs, n_pairs, read_len=150, ins_lo=250, ins_hi=450,
               sub_rate=0.005, seed=1):
    """Paired reads: mate 1 forward, mate 2 reverse-complement of the
    fragment's far end; insert U[ins_lo, ins_hi] clipped to the gene."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(s) for s in seqs])
    ok = np.nonzero(lens >= read_len)[0]
    g = ok[rng.integers(0, len(ok), size=n_pairs)]
    m1 = np.empty((n_pairs, read_len), np.uint8)
    m2 = np.empty((n_pairs, read_len), np.uint8)
    for i in range(n_pairs):
        s = seqs[g[i]]
        ins = min(int(rng.integers(ins_lo, ins_hi + 1)), len(s))
        st = int(rng.integers(0, len(s) - ins + 1))
        frag = s[st:st + ins]
        a = frag[:read_len].copy()
        b = revcomp_codes(frag[-read_len:]).copy()
        if rng.random() < 0.5:  # sequenced from the other strand
            a, b = b, a
        m1[i], m2[i] = a, b
    for m in (m1, m2):
        x = rng.random(m.shape) < sub_rate
        m[x] = (m[x] + rng.integers(1, 4, size=int(x.sum()), dtype=np.uint8)) & 3
    return m1, m2, g


def make_long_reads(genome, n_reads, read_len=10000, sub=0.04, dele=0.03,
                    ins=0.03, rc_frac=0.5, seed=1):
    """ONT-like reads from one genome (list of uint8 arrays, ragged)."""
    rng = np.random.default_rng(seed)
    out = []
    G = len(genome)
    for _ in range(n_reads):
        st = int(rng.integers(0, G - read_len + 1))
        w = genome[st:st + read_len].copy()
        m = rng.random(read_len) < sub
        w[m] = (w[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        keep = rng.random(read_len) >= dele
        w = w[keep]
        ipos = np.nonzero(rng.random(len(w)) < ins)[0]
        if len(ipos):
            w = np.insert(w, ipos, rng.integers(0, 4, size=len(ipos), dtype=np.uint8))
        if rng.random() < rc_frac:
            w = revcomp_codes(w)
        out.append(np.ascontiguousarray(w))
    return out
