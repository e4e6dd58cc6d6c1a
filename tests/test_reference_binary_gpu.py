"""End to end against the compiled reference itself (oracle/_ref/kma, which travels to the GPU box): a seeded read set with
substitutions, indels, unmappable reads, partly foreign reads, N's and ragged lengths goes through `kma -1t1 -t 1` and through
examples/kmahip_map (kmahip_ingest_*, kmahip_run_se, the writers); `.res`, `.fsa`, `.aln` and `.frag.gz` must be identical."""
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


@pytest.mark.parametrize("seed,families,variants,mf", [(1, 40, 5, None), (2, 12, 12, None), (3, 20, 5, 1000)])
def test_whole_run_equals_reference_binary(tmp_path, seed, families, variants, mf):
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
    extra = ["-mf", str(mf)] if mf else []        # (-mf: fragments per assembly chunk, 60 chunks here instead of one)
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-t", "1"] + extra, check=True, stderr=subprocess.DEVNULL)
    # (the row order of the .frag.gz is worked out by several threads from 64 k fragments per thread on; seeds 2 and 3 make it 500)
    env = dict(os.environ, KMAHIP_ROW_GRAIN="500") if seed > 1 else None
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-1t1"] + extra, check=True,
                   stderr=subprocess.DEVNULL, env=env)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 40000


@pytest.mark.parametrize("mf,apm", [(None, ["-apm", "p"]), (1001, ["-apm", "p"]), (7, ["-apm", "p"]), (None, ["-apm", "u"]), (1001, []), (7, ["-apm", "u"])])
def test_whole_paired_run_equals_reference_binary(tmp_path, mf, apm):
    """`-ipe r1 r2 -apm p -1t1 -t 1` -- and `-apm u`, and no -apm at all, which is the union pairing too (kma.c:206: save_kmers_unionPair,
    alnFragsUnionPE) --: pairs with substitutions, some mates foreign or too short after trimming (single records in the
    pair stream), some with an insertion or deletion, through the reference and through examples/kmahip_map -ipe. With -mf 1001
    the assembly chunks close after whole records: couples straddle the limit (chunks of 1002 fragments, conclave.c:164-196); -mf 7
    makes eight thousand chunks, half of them closed by a couple."""
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
    for i in rng.choice(len(r1), 2500, replace=False):                   # an insertion or a deletion in a mate (the pile-up order matters)
        r = r1 if rng.random() < 0.5 else r2
        a = int(rng.integers(30, 120))
        r[i] = np.concatenate([r[i][:a], rng.integers(0, 4, 2, dtype=np.uint8), r[i][a:148]]) if rng.random() < 0.5 else np.concatenate([r[i][:a], r[i][a + 2:], r[i][:2]])
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    extra = ["-mf", str(mf)] if mf else []
    for path, rs, qs, tag in ((tmp_path / "r1.fq", r1, q1, b"/1"), (tmp_path / "r2.fq", r2, q2, b"/2")):
        with open(path, "wb") as f:
            for i, (r, q) in enumerate(zip(rs, qs)):
                f.write(b"@p%d" % i + tag + b"\n" + lut[r].tobytes() + b"\n+\n" + q + b"\n")
    subprocess.run([KMA, "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-t", "1"] + apm + extra,
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-t_db", prefix, "-o", str(tmp_path / "got"), "-1t1"] + apm + extra,
                   check=True, stderr=subprocess.DEVNULL, env=dict(os.environ, KMAHIP_ROW_GRAIN="700") if mf else None)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 50000


def _fuzz_seeds():
    e = os.environ.get("KMA_PE_FUZZ_SEEDS")          # "a:b" = range(a, b); the suite runs three
    if e:
        a, b = e.split(":")
        return list(range(int(a), int(b)))
    return [1, 2, 3]


def _apm_modes():
    return ["p", "u"] if not os.environ.get("KMA_PE_FUZZ_APM") else os.environ["KMA_PE_FUZZ_APM"].split(",")


@pytest.mark.parametrize("apm", _apm_modes())
@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_paired_fuzz_equals_reference_binary(tmp_path, seed, apm):
    """a random database and pair set per seed: mates of ragged lengths with indels, foreign ends, N's, unmappable mates, mates the
    quality trim removes or shortens, mates drawn from different variants of a family or from different families; maxFrag drawn
    too. `kma -ipe r1 r2 -apm p|u -1t1 -t 1` against examples/kmahip_map -ipe: .res, .fsa, .frag.gz."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(1000 + seed)
    fam, var = int(rng.integers(3, 30)), int(rng.integers(2, 9))
    names, seqs = synth.make_gene_db(fam, var, 400, 1600, float(rng.choice([0.01, 0.03, 0.06])), seed=2000 + seed)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    n = 6000
    a, b = _reads(seqs, n, rng), _reads(seqs, n, rng)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(tmp_path / "r1.fq", "wb") as f1, open(tmp_path / "r2.fq", "wb") as f2:
        for i in range(n):
            u = rng.random()
            if u < 0.7:                                   # a proper couple: mate 2 from the same template as mate 1, other strand, nearby
                s = seqs[int(rng.integers(0, len(seqs)))]
                L1, L2 = int(rng.integers(40, 200)), int(rng.integers(40, 200))
                ins = int(rng.integers(max(L1, L2), max(L1, L2) + 300))
                st = int(rng.integers(0, max(1, len(s) - ins)))
                frag = s[st:st + ins]
                m1, m2 = frag[:L1].copy(), synth.revcomp_codes(frag[-L2:]).copy()
                for m in (m1, m2):
                    x = rng.random(len(m)) < 0.01
                    m[x] = (m[x] + rng.integers(1, 4, int(x.sum()), dtype=np.uint8)) & 3
                if rng.random() < 0.5:
                    m1, m2 = m2, m1
            elif u < 0.85:
                m1, m2 = a[i], b[i]                        # two unrelated reads (with all of _reads' oddities)
            else:
                m1, m2 = a[i], synth.revcomp_codes(a[i])[:max(20, len(a[i]) // 2)].copy()      # overlapping mates
            q1, q2 = bytearray(b"I" * len(m1)), bytearray(b"I" * len(m2))
            v = rng.random()
            if v < 0.05:
                q1[8:] = b"#" * (len(q1) - 8)             # trimmed below the minimum length: mate 2 goes on as a single
            elif v < 0.1:
                q2[8:] = b"#" * (len(q2) - 8)
            elif v < 0.15:
                q1[len(q1) // 2:] = b"#" * (len(q1) - len(q1) // 2)      # shortened
            f1.write(b"@p%d/1\n" % i + lut[m1].tobytes() + b"\n+\n" + bytes(q1) + b"\n")
            f2.write(b"@p%d/2\n" % i + lut[m2].tobytes() + b"\n+\n" + bytes(q2) + b"\n")
    extra = [] if seed % 3 == 0 else ["-mf", str(int(rng.integers(2, 3000)))]
    subprocess.run([KMA, "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-apm", apm, "-t", "1"] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-t_db", prefix, "-o", str(tmp_path / "got"), "-1t1", "-apm", apm] + extra,
                   check=True, stderr=subprocess.DEVNULL, env=dict(os.environ, KMAHIP_ROW_GRAIN="300"))
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 3000


def test_paired_stream_of_singles_equals_reference_binary(tmp_path):
    """`-ipe` where one mate of EVERY pair falls to the quality trim: the pair stream holds single records only (no pair batch at all
    in kmahip_run_pe), and a second run where both mates of every pair are foreign (nothing maps: empty .res, no rows)."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(15)
    names, seqs = synth.make_gene_db(20, 5, 700, 1400, 0.04, seed=78)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    m1, m2, _ = synth.make_pairs(seqs, 4000, seed=19)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    bad = b"I" * 10 + b"#" * 140
    for case in ("singles", "foreign"):
        for path, ms, tag, other in ((tmp_path / "r1.fq", m1, b"/1", 0), (tmp_path / "r2.fq", m2, b"/2", 1)):
            with open(path, "wb") as f:
                for i, r in enumerate(ms):
                    if case == "foreign":
                        r = rng.integers(0, 4, 150, dtype=np.uint8)
                    q = bad if case == "singles" and (i & 1) == other else b"I" * 150
                    f.write(b"@p%d" % i + tag + b"\n" + lut[r].tobytes() + b"\n+\n" + q + b"\n")
        subprocess.run([KMA, "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-apm", "p", "-t", "1"],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-t_db", prefix, "-o", str(tmp_path / "got"), "-1t1", "-apm", "p"],
                       check=True, stderr=subprocess.DEVNULL)
        assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read(), case
        assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read(), case
        assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read(), case
        got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
        assert got == ref, case
        assert (got.count(b"\n") > 3000) == (case == "singles")


@pytest.mark.parametrize("mode", ["-1t1", "default"])
def test_run_where_nothing_maps_equals_reference_binary(tmp_path, mode):
    """reads foreign to the database: header-only .res, empty .fsa, a .frag.gz that inflates to nothing -- all three files written"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(16)
    names, seqs = synth.make_gene_db(10, 5, 700, 1400, 0.04, seed=79)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(tmp_path / "r.fq", "wb") as f:
        for i in range(2000):
            f.write(b"@r%d\n" % i + lut[rng.integers(0, 4, 150, dtype=np.uint8)].tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
    for o in ("got.frag.gz", "ref.frag.gz"):
        with open(tmp_path / o, "wb") as f:
            f.write(b"stale")
    subprocess.run([KMA, "-i", str(tmp_path / "r.fq"), "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"] + (["-1t1"] if mode == "-1t1" else []),
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", str(tmp_path / "r.fq"), "-t_db", prefix, "-o", str(tmp_path / "got"),
                    "-1t1" if mode == "-1t1" else "-chain"], check=True, stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    assert gzip.open(tmp_path / "got.frag.gz", "rb").read() == gzip.open(tmp_path / "ref.frag.gz", "rb").read() == b""


@pytest.mark.parametrize("mode", ["default", "-1t1"])
def test_bcnano_without_mt1_equals_reference_binary(tmp_path, mode):
    """`-bcNano` without `-Mt1` (the common way to run ONT reads): nanoCaller + significantAnd90Nuc in the consensus of the default mode
    and of `-1t1`; long reads (2-6 kb, 10 % errors) whose default-mode records carry query bounds through the long-read traceback."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(31)
    genes = [rng.integers(0, 4, int(rng.integers(3000, 9000)), dtype=np.uint8) for _ in range(12)]
    genes += [g.copy() for g in genes[:4]]
    for g in genes[12:]:                                   # four near-copies: ties and template choice
        x = rng.random(len(g)) < 0.02
        g[x] = (g[x] + rng.integers(1, 4, int(x.sum()), dtype=np.uint8)) & 3
    names = [f"g{i}" for i in range(len(genes))]
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, genes)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = []
    for i in range(600):
        g = genes[int(rng.integers(0, len(genes)))]
        r = synth.make_long_reads(g, 1, read_len=int(rng.integers(2000, min(6000, len(g)))), seed=1000 + i)[0]
        if i % 9 == 0:                                     # a chimera: two genes in one read
            h = genes[int(rng.integers(0, len(genes)))]
            r = np.concatenate([r, synth.make_long_reads(h, 1, read_len=2000, seed=5000 + i)[0]])
        reads.append(r)
    fq = str(tmp_path / "ont.fq")
    synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    flags = ["-1t1"] if mode == "-1t1" else []
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-bcNano", "-t", "1"] + flags, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-bcNano"] + (flags or ["-chain"]),
                   check=True, stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    fsa = open(tmp_path / "got.fsa", "rb").read()
    assert fsa == open(tmp_path / "ref.fsa", "rb").read() and any(c in fsa for c in b"acgt")      # (lower case: what nanoCaller leaves where depth is thin)
    got = gzip.open(tmp_path / "got.frag.gz").read()
    assert got == gzip.open(tmp_path / "ref.frag.gz").read() and got.count(b"\n") > 500


@pytest.mark.parametrize("mode", ["-1t1", "default", "-ipe", "-Mt1", "-Mt1 in one call"])
def test_empty_input_equals_reference_binary(tmp_path, mode):
    """a FASTQ file without a record (and one whose only record is too short to keep): the three outputs as the reference writes them"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    names, seqs = synth.make_gene_db(5, 3, 700, 900, 0.04, seed=80)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    for case, text in (("empty", b""), ("short", b"@r0\nACGTACGT\n+\nIIIIIIII\n")):
        for f in ("r1.fq", "r2.fq"):
            (tmp_path / f).write_bytes(text)
        if mode == "-ipe":
            ra = ["-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-apm", "p", "-1t1"]
            ga = ["-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-1t1", "-apm", "p"]
        elif mode.startswith("-Mt1"):
            ra = ga = ["-i", str(tmp_path / "r1.fq"), "-Mt1", "2", "-bcNano"]
        else:
            ra = ["-i", str(tmp_path / "r1.fq")] + (["-1t1"] if mode == "-1t1" else [])
            ga = ["-i", str(tmp_path / "r1.fq"), "-1t1" if mode == "-1t1" else "-chain"]
        r = subprocess.run([KMA] + ra + ["-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        g = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map")] + ga + ["-t_db", prefix, "-o", str(tmp_path / "got")], stderr=subprocess.PIPE,
                           env=dict(os.environ, **({"KMAHIP_MAP_ONE_BATCH": "1"} if mode.endswith("one call") else {})))
        assert g.returncode == 0, (case, g.stderr.decode()[-300:])
        if r.returncode == 0 and os.path.exists(tmp_path / "ref.res"):
            assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read(), case
            assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read(), case
            assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read(), case
            assert gzip.open(tmp_path / "got.frag.gz", "rb").read() == gzip.open(tmp_path / "ref.frag.gz", "rb").read() == b"", case
        for f in ("ref.res", "ref.fsa", "ref.aln", "ref.frag.gz", "got.res", "got.fsa", "got.aln", "got.frag.gz"):
            if os.path.exists(tmp_path / f):
                os.unlink(tmp_path / f)


@pytest.mark.parametrize("env", [{}, {"KMAHIP_MAP_BATCH": "97"}, {"KMAHIP_MAP_BATCH_BASES": "300000", "KMAHIP_INGEST_REGION": "4096"}, {"KMAHIP_MAP_ONE_BATCH": "1"}],
                         ids=["session", "batches_of_97_reads", "batches_by_bases", "one_batch"])
def test_whole_mt1_bcnano_run_equals_reference_binary(tmp_path, env):
    """BASELINE config C4 through the C host program: `kma -i ont.fq -t_db db -Mt1 1 -bcNano -t 1` vs `kmahip_map ... -Mt1 1 -bcNano`
    on ONT-like reads (2-12 kb, 10 % errors, both strands, some with foreign chunks, N's, low-quality ends that the trim removes,
    unmappable reads) against one 400 kb genome with a few repeats, indexed by the reference's own `kma index`. The host program feeds
    the reads batch by batch (kmahip_session_set_mt1: every batch traced as it comes): in one batch, in batches of 97 reads, in batches
    closed by their bases, and through the one-call path (kmahip_run_mt1)."""
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
    run = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-Mt1", "1", "-bcNano"], check=True,
                         stderr=subprocess.PIPE, env=dict(os.environ, **env))
    if "KMAHIP_MAP_BATCH" in env or "KMAHIP_MAP_BATCH_BASES" in env:
        import re
        m = re.search(rb"in (\d+) batches", run.stderr)
        assert m and int(m.group(1)) >= 4, run.stderr[-400:]
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref
    assert got.count(b"\n") > 1000


def test_deep_pile_up_of_long_reads_equals_reference_binary(tmp_path):
    """The pile-up at the depth BASELINE config C4 has at its own size (1 M reads of 10 kb on 5 Mb: 2 000 x), which no other test reaches:
    5 200 ONT-like reads of 10 kb (10 % errors: every read brings hundreds of insertion runs) on ONE template of 45 kb -- depth above
    1 000 x everywhere, insertion chains of dozens of columns between neighbouring template positions whose order and starting depths
    follow the order of the reads, and base counts that pass the 16-bit saturation of alnToMat's counters with -Mt1's raw piling
    (assembly.c:1359-1442) -- `kma -Mt1 1 -bcNano -t 1` against `kmahip_map -Mt1 1 -bcNano`: .res, .fsa, .frag.gz."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(23)
    genome = rng.integers(0, 4, 45_000, dtype=np.uint8)
    fsa = str(tmp_path / "g.fsa")
    synth.write_fasta(fsa, ["deep template"], [genome])
    prefix = str(tmp_path / "db")
    subprocess.run([KMA, "index", "-i", fsa, "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_long_reads(genome, 5200, read_len=10000, sub=0.04, dele=0.03, ins=0.03, seed=6)
    fq = str(tmp_path / "ont.fq")
    synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-Mt1", "1", "-bcNano", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-Mt1", "1", "-bcNano"], check=True,
                   stderr=subprocess.PIPE)
    res = open(tmp_path / "ref.res").read().splitlines()
    assert len(res) == 2 and float(res[1].split("\t")[8]) > 1000.0, res          # the Depth column
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    assert gzip.open(tmp_path / "got.frag.gz", "rb").read() == gzip.open(tmp_path / "ref.frag.gz", "rb").read()


@pytest.mark.parametrize("ts", [2, 7, 30])
def test_seed_trimming_equals_reference_binary(tmp_path, ts):
    """`-ts n` (trimSeeds, chain.c:493-528, called by KMA() align.c:413 -- not by KMA_score): the front of every seed of the best chain goes
    back to the DP problem before it, in the short-read traceback and in the long-read pipeline (`-ts 2` is part of the reference's -ont
    preset). Reads with substitutions and indels (several seeds per read, seeds shorter than the trim), both modes, and long reads with
    -Mt1: .res, .fsa, .aln and .frag.gz against the binary."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(100 + ts)
    names, seqs = synth.make_gene_db(n_families=25, variants=3, len_lo=900, len_hi=2500, seed=5)
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads, *_ = synth.make_reads(seqs, 2500, read_len=150, sub_rate=0.03, seed=3)
    reads = list(reads)
    cat = np.concatenate(seqs)
    reads += synth.make_long_reads(cat, 500, read_len=250, sub=0.02, dele=0.015, ins=0.015, seed=4)
    for g in rng.integers(0, len(seqs), 12):
        reads += synth.make_long_reads(seqs[g], 6, read_len=min(2000, len(seqs[g]) - 10), sub=0.04, dele=0.03, ins=0.03, seed=int(g))
    reads = [reads[i] for i in rng.permutation(len(reads))]
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads, prefix="q")
    for mode in (["-1t1"], [], ["-Mt1", "4", "-bcNano"]):
        for f in ("ref", "got"):
            for ext in (".res", ".fsa", ".aln", ".frag.gz"):
                if os.path.exists(tmp_path / (f + ext)):
                    os.unlink(tmp_path / (f + ext))
        subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1", "-ts", str(ts)] + mode, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-ts", str(ts)] + mode, check=True, stderr=subprocess.PIPE)
        assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read(), mode
        assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read(), mode
        assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read(), mode
        assert gzip.open(tmp_path / "got.frag.gz", "rb").read() == gzip.open(tmp_path / "ref.frag.gz", "rb").read(), mode
    if ts == 30:
        # a trim of 30 changes alignments on this input (giving two exactly matching bases back to a DP problem changes nothing, as a
        # rule): the comparison above would not pass with the option ignored
        subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "plain"), "-Mt1", "4", "-bcNano"], check=True, stderr=subprocess.PIPE)
        assert gzip.open(tmp_path / "plain.frag.gz", "rb").read() != gzip.open(tmp_path / "got.frag.gz", "rb").read()


def _pe_branch_db_and_pairs(rng, n_fam, n):
    """A database and pairs built to leave the proper-pair branch of alnFragsPenaltyPE (alnfrags.c:1777-1970). Every family has two
    templates, X and Y = X with a substitution every `gap` bases. A pair takes mate 1 from X and mate 2 from Y, placed so that the
    substitutions fall mirror-symmetrically in the two mates: on X mate 1 is perfect and mate 2 carries c substitutions, on Y the other
    way round -- the two templates tie in stage 2 (a couple with both in its list), and in stage 3a each mate scores best on its own
    template: best + best_r exceeds every single template's joint score by 3 c, beyond PE = 7 once c >= 3 -> an unmated pair."""
    names, seqs, pairs1, pairs2 = [], [], [], []
    meta = []
    for f in range(n_fam):
        L = int(rng.integers(800, 1500))
        gap = int(rng.choice([30, 40, 50]))
        a = int(rng.integers(0, gap))
        x = rng.integers(0, 4, L, dtype=np.uint8)
        y = x.copy()
        pos = np.arange(a, L, gap)
        y[pos] = (y[pos] + rng.integers(1, 4, len(pos), dtype=np.uint8)) & 3
        names += [f"fam{f}_X", f"fam{f}_Y"]
        seqs += [x, y]
        meta.append((L, gap, a))
    for i in range(n):
        f = int(rng.integers(0, n_fam))
        L, gap, a = meta[f]
        x, y = seqs[2 * f], seqs[2 * f + 1]
        st = int(rng.integers(0, L - 400))
        # mate 2 window [st2, st2 + 150): st + st2 = 2 a - 149 (mod gap) mirrors the substitution pattern
        st2 = st + 100 + int(rng.integers(0, 150))
        u = rng.random()
        if u < 0.7:
            st2 += (2 * a - 149 - st - st2) % gap
        st2 = min(st2, L - 150)
        m1, m2 = x[st:st + 150].copy(), synth.revcomp_codes(y[st2:st2 + 150])
        if u >= 0.85:                                   # plain pairs, some with errors, to keep the other branches in the mix
            m2 = synth.revcomp_codes(x[st2:st2 + 150])
            e = rng.random(150) < 0.02
            m2 = m2.copy(); m2[e] = (m2[e] + 1) & 3
        if rng.random() < 0.5:
            m1, m2 = m2, m1
        pairs1.append(m1); pairs2.append(np.ascontiguousarray(m2))
    return names, seqs, np.array(pairs1), np.array(pairs2)


def test_pe_unmated_and_single_mate_branches_equal_reference_binary(tmp_path):
    """The branch of alnFragsPenaltyPE no earlier fixture reached: a couple whose mates score best on different templates (unmated
    pair: update_Scores_se twice, alnfrags.c:1820-1891). The pairs are built for that; our own pe_kind says the branch is taken
    by the hundred; `.res`, consensus and `.frag.gz` (every fragment's orientation, score, start, end and template) must equal the
    reference's. The two single-record branches (:1892-1970) cannot be reached by any input: with one of best_read_score /
    best_read_score_r zero, compScore (the maximum over templates of mate 1's kept score + mate 2's raw score, :1771-1775) is at
    least the other one, so `score <= compScore + PE` (:1790) always takes the proper-pair branch first."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    from kma_amd import binding
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(21)
    names, seqs, m1, m2 = _pe_branch_db_and_pairs(rng, 60, 30000)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    inter = np.empty((2 * len(m1), 150), np.uint8)
    inter[0::2], inter[1::2] = m1, m2
    db = binding.KmaHipDB(prefix)
    try:
        _, h = db.map_pe(formats.pack_fixed(inter))
    finally:
        db.close()
    kinds = np.bincount(h["kind"], minlength=5)
    assert kinds[1] > 1000 and kinds[2] > 100 and kinds[3] == 0 and kinds[4] == 0, kinds
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for path, rs, tag in ((tmp_path / "r1.fq", m1, b"/1"), (tmp_path / "r2.fq", m2, b"/2")):
        with open(path, "wb") as f:
            for i, r in enumerate(rs):
                f.write(b"@p%d" % i + tag + b"\n" + lut[r].tobytes() + b"\n+\n" + b"I" * 150 + b"\n")
    subprocess.run([KMA, "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-apm", "p", "-t", "1"],
                   check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-t_db", prefix, "-o", str(tmp_path / "got"), "-1t1", "-apm", "p"],
                   check=True, stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref


@pytest.mark.parametrize("k,flags", [(12, ["-ME"]), (9, [])])
def test_direct_address_index_written_by_kma_index(tmp_path, k, flags):
    """An index in the reference's direct-address form (`kma index -ME`, hashmapkma.c:264-273, 777-812; without the flag the
    builder switches to it by itself once the table would be half of 4^k, hashmap.c:204-209 -- k = 9 here): exist[] holds a
    value-list offset per possible k-mer and there are no key arrays. The whole run against it must equal the reference's."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(40 + k)
    names, seqs = synth.make_gene_db(30, 4, 500, 1100, 0.04, seed=300 + k)
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix, "-k", str(k)] + flags, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    hdr = np.fromfile(prefix + ".comp.b", dtype=np.uint8, count=52)
    size = int(hdr[20:28].view(np.uint64)[0])
    assert size == 4 ** k, "the index is not in direct-address form"
    reads = _reads(seqs, 20000, rng)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, reads, lens=None)
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-1t1", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-1t1"], check=True,
                   stderr=subprocess.DEVNULL)
    assert open(tmp_path / "got.res", "rb").read() == open(tmp_path / "ref.res", "rb").read()
    assert open(tmp_path / "got.fsa", "rb").read() == open(tmp_path / "ref.fsa", "rb").read()
    assert open(tmp_path / "got.aln", "rb").read() == open(tmp_path / "ref.aln", "rb").read()
    got, ref = gzip.open(tmp_path / "got.frag.gz", "rb").read(), gzip.open(tmp_path / "ref.frag.gz", "rb").read()
    assert got == ref and got.count(b"\n") > 10000
