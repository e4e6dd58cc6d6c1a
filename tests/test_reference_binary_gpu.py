"""End to end against the compiled reference itself (oracle/_ref/kma, which travels to the GPU box): a seeded read set with
substitutions, indels, unmappable reads, partly foreign reads, N's and ragged lengths goes through `kma -1t1 -t 1` and through
examples/kmahip_map (kmahip_ingest_*, kmahip_run_se, the writers); `.res`, `.fsa` and `.frag.gz` must be identical."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from kma_amd import formats, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")


def _reads(seqs, n, rng):
    out = []
    for i in range(n):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(60, 251))
        a = int(rng.integers(0, max(1, len(s) - L)))
        r = s[a:a + L].copy()
        u = rng.random()
        if u < 0.03:                                   # unmappable
            r = rng.integers(0, 4, L, dtype=np.uint8)
        elif u < 0.08:                                 # foreign end (either side)
            j = rng.integers(0, 4, int(rng.integers(20, 140)), dtype=np.uint8)
            r = np.concatenate([r, j]) if rng.random() < 0.5 else np.concatenate([j, r])
        elif u < 0.16:                                 # an indel or two
            parts, p = [], 0
            while p < len(r):
                e = min(len(r), p + int(rng.integers(25, 110)))
                parts.append(r[p:e])
                v = rng.random()
                if v < 0.45:
                    parts.append(rng.integers(0, 4, int(rng.integers(1, 5)), dtype=np.uint8))
                elif v < 0.9:
                    e = min(len(r), e + int(rng.integers(1, 5)))
                p = e
            r = np.concatenate(parts)
        x = rng.random(len(r)) < 0.008
        r = r.copy()
        r[x] = (r[x] + rng.integers(1, 4, int(x.sum()), dtype=np.uint8)) & 3
        if rng.random() < 0.03:
            r[int(rng.integers(0, len(r)))] = 4        # an N
        if rng.random() < 0.5:
            r = synth.revcomp_codes(r)
        out.append(np.ascontiguousarray(r.astype(np.uint8)))
    return out


@pytest.mark.parametrize("seed,families,variants", [(1, 40, 5), (2, 12, 12)])
def test_whole_run_equals_reference_binary(tmp_path, seed, families, variants):
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(seed)
    names, seqs = synth.make_gene_db(families, variants, 500, 1300, 0.04, seed=100 + seed)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads = _reads(seqs, 60000, rng)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, reads, lens=None)
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got")], check=True,
                   stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 40000


def test_whole_paired_run_equals_reference_binary(tmp_path):
    """`-ipe r1 r2 -apm p -1t1 -t 1`: pairs with substitutions, some mates foreign or too short after trimming (single records in the
    pair stream), through the reference and through examples/kmahip_map -ipe."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(5)
    names, seqs = synth.make_gene_db(30, 5, 700, 1400, 0.04, seed=77)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    m1, m2, _ = synth.make_pairs(seqs, 30000, seed=9)
    r1, r2 = [r.copy() for r in m1], [r.copy() for r in m2]
    q1, q2 = [b"I" * 150] * len(r1), [b"I" * 150] * len(r2)
    for i in rng.choice(len(r1), 1500, replace=False):                   # a foreign mate
        (r1 if rng.random() < 0.5 else r2)[i] = rng.integers(0, 4, 150, dtype=np.uint8)
    for i in rng.choice(len(r1), 1500, replace=False):                   # a mate that the quality trim shortens below -ml
        q = bytearray(b"I" * 150)
        q[10:] = b"#" * 140
        if rng.random() < 0.5:
            q1[i] = bytes(q)
        else:
            q2[i] = bytes(q)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for path, rs, qs, tag in ((tmp_path / "r1.fq", r1, q1, b"/1"), (tmp_path / "r2.fq", r2, q2, b"/2")):
        with open(path, "wb") as f:
            for i, (r, q) in enumerate(zip(rs, qs)):
                f.write(b"@p%d" % i + tag + b"\n" + lut[r].tobytes() + b"\n+\n" + q + b"\n")
    subprocess.run([KMA, "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-apm", "p", "-t", "1"],
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-t_db", prefix, "-o", str(tmp_path / "got")],
                   check=True, stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 50000


def test_whole_mt1_bcnano_run_equals_reference_binary(tmp_path):
    """BASELINE config C4 through the C host program: `kma -i ont.fq -t_db db -Mt1 1 -bcNano -t 1` vs `kmahip_map ... -Mt1 1 -bcNano`
    on ONT-like reads (2-12 kb, 10 % errors, both strands, some with foreign chunks, N's, low-quality ends that the trim removes,
    unmappable reads) against one 400 kb genome with a few repeats, indexed by the reference's own `kma index`."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(11)
    G = 400000
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    for _ in range(6):                                   # repeats: direct and inverted copies of 300-1500 bases
        L = int(rng.integers(300, 1500)); a = int(rng.integers(0, G - L)); b = int(rng.integers(0, G - L))
        seg = genome[a:a + L].copy()
        genome[b:b + L] = synth.revcomp_codes(seg) if rng.random() < 0.5 else seg
    fsa = str(tmp_path / "g.fsa")
    synth.write_fasta(fsa, ["chr test genome"], [genome])
    prefix = str(tmp_path / "db")
    subprocess.run([KMA, "index", "-i", fsa, "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = []
    for L, n, seed in ((2000, 500, 1), (5000, 400, 2), (12000, 120, 3), (300, 200, 4)):
        reads += synth.make_long_reads(genome, n, read_len=L, sub=0.04, dele=0.03, ins=0.03, seed=seed)
    quals = []
    for i, r in enumerate(reads):
        u = rng.random()
        if u < 0.03:
            reads[i] = r = rng.integers(0, 4, len(r), dtype=np.uint8)
        elif u < 0.08:
            p = int(rng.integers(0, len(r)))
            reads[i] = r = np.concatenate([r[:p], rng.integers(0, 4, int(rng.integers(40, 400)), dtype=np.uint8), r[p:]])
        elif u < 0.12:
            r = r.copy(); r[rng.integers(0, len(r), int(rng.integers(1, 6)))] = 4; reads[i] = r
        q = bytearray(b"5" * len(r))
        if rng.random() < 0.1:                           # a low-quality end: trimmed by stage 1 (-mp 20)
            k = int(rng.integers(1, 60)); q[-k:] = b"#" * k
        quals.append(bytes(q))
    order = rng.permutation(len(reads))
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    fq = str(tmp_path / "ont.fq")
    with open(fq, "wb") as f:
        for j, i in enumerate(order):
            f.write(b"@ont%d some comment\n" % j + lut[reads[i]].tobytes() + b"\n+\n" + quals[i] + b"\n")
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-Mt1", "1", "-bcNano", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-Mt1", "1", "-bcNano"], check=True,
                   stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 1000
