"""Stage 2 of KMA's default mode (no -1t1) on the HIP path, kmahip_scan_chain: the S2 records against the reference's own
`-s2` tap of the committed fixtures, against the oracle, and against the compiled reference on chimeric reads (several chains per read)."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_util
import oracle
from kma_amd import binding, formats, synth
from test_oracle_golden import _chimeric_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")


def _records(o):
    return [(int(o["read"][x]), int(o["rc_flag"][x]), int(o["emit_rc"][x]), int(o["q_start"][x]), int(o["q_end"][x]),
             tuple(int(t) for t in o["T"][o["T_off"][x]:o["T_off"][x + 1]])) for x in range(len(o["read"]))]


@pytest.mark.parametrize("name", ["se", "long"])
def test_chain_records_equal_reference_tap_and_oracle(tmp_path, name):
    g = golden_util.load_se(tmp_path, name)
    tap, _ = formats.parse_s2(gzip.open(os.path.join(g["dir"], "s2_chain.bin.gz")).read())
    b = g["batch"]
    db = binding.KmaHipDB(g["prefix"])
    try:
        got = _records(db.scan_chain(b))
    finally:
        db.close()
    odb = oracle.OracleDB(g["prefix"])
    want = [(i, rf, er, qs, qe, tuple(int(t) for t in T)) for i, recs in enumerate(odb.scan_chain(b)) for rf, er, qs, qe, T in recs]
    assert got == want and len(got) > 200
    # ... and the reference's tap, read by read (reads with an N among their first k - 1 bases left out: tests/test_oracle_golden.py)
    by_read = {}
    for w in tap:
        by_read.setdefault(w["hdr"][:len(w["hdr"]) - 9], []).append(w)
    mine = {}
    for rec in got:
        mine.setdefault(rec[0], []).append(rec)
    checked = 0
    for i in range(b.n):
        Ni = b.N[b.N_off[i]:b.N_off[i + 1]]
        if len(Ni) and int(Ni[0]) < 15:
            continue
        hdr = g["s1"][i]["hdr"]
        ws, rs = by_read.get(hdr, []), mine.get(i, [])
        assert len(ws) == len(rs), (i, hdr)
        for w, (_, rf, er, qs, qe, T) in zip(ws, rs):
            checked += 1
            assert w["hdr"] == hdr + b"\x00" + struct.pack("<2i", qs, qe) and w["rc_flag"] == rf and tuple(int(t) for t in w["T"]) == T
            L = int(b.length[i])
            seq = b.seq[b.seq_off[i]:b.seq_off[i] + ((L + 31) >> 5)]
            if er:
                seq, _ = oracle.rc_packed(seq, L, Ni)
            assert np.array_equal(w["seq"], seq)
    assert checked > 200


@pytest.mark.parametrize("seed", [1, 2])
def test_chain_records_equal_reference_binary_on_chimeric_reads(tmp_path, seed):
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    rng = np.random.default_rng(300 + seed)
    names, seqs = synth.make_gene_db(60, 5, 300, 900, 0.05, seed=400 + seed)
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = _chimeric_reads(seqs, 20000, rng)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads)
    tap = subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "o"), "-t_db", prefix, "-t", "1", "-s2"], check=True, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL).stdout
    want, _ = formats.parse_s2(tap)
    b = formats.pack_ragged(reads)
    db = binding.KmaHipDB(prefix)
    try:
        got = _records(db.scan_chain(b))
    finally:
        db.close()
    flat = [(b"r%d" % r + bytes(2) + struct.pack("<2i", qs, qe), rf, T) for r, rf, er, qs, qe, T in got]
    ref = [(w["hdr"], w["rc_flag"], tuple(int(x) for x in w["T"])) for w in want]
    assert sum(1 for x in range(1, len(got)) if got[x][0] == got[x - 1][0]) > 2000          # many reads map in pieces
    for x, (a, c) in enumerate(zip(flat, ref)):
        assert a == c, (x, a, c)
    assert len(flat) == len(ref)


def _run_both(tmp_path, prefix, fq, extra=(), env=None):
    """examples/kmahip_map in the default mode: through the batched session (kmahip_session_set_chain) unless env says otherwise"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got"), "-chain"] + list(extra), check=True,
                   stderr=subprocess.DEVNULL, env=dict(os.environ, KMAHIP_ROW_GRAIN="700", **(env or {})))
    return [open(tmp_path / "got.res", "rb").read(), open(tmp_path / "got.fsa", "rb").read(), gzip.open(tmp_path / "got.frag.gz").read()]


@pytest.mark.parametrize("name", ["se", "long"])
def test_whole_default_mode_run_equals_reference_files(tmp_path, name):
    """examples/kmahip_map -chain (kmahip_run_chain: scan_chain, then stage 3a / ConClave / traceback / pile-up with the records' query
    bounds) against the files the reference wrote without -1t1 for the same reads (tests/golden/make_golden_chain.py)."""
    import shutil
    g = golden_util.load_se(tmp_path, name)
    fq = str(tmp_path / "reads.fq")
    with gzip.open(os.path.join(g["dir"], "reads.fq.gz"), "rb") as f, open(fq, "wb") as o:
        shutil.copyfileobj(f, o)
    res, fsa, frag = _run_both(tmp_path, g["prefix"], fq)
    assert res == open(os.path.join(g["dir"], "chain.res"), "rb").read()
    assert fsa == gzip.open(os.path.join(g["dir"], "chain.fsa.gz")).read()
    assert frag == gzip.open(os.path.join(g["dir"], "chain.frag.gz")).read()


def _chain_fuzz_seeds():
    e = os.environ.get("KMA_CHAIN_FUZZ_SEEDS")       # "a:b" = range(a, b); the suite runs 1 and 2
    if e:
        a, b = e.split(":")
        return list(range(int(a), int(b)))
    return [1, 2]


@pytest.mark.parametrize("seed", _chain_fuzz_seeds())
def test_whole_default_mode_run_equals_reference_binary_on_chimeric_reads(tmp_path, seed):
    """reads that map in pieces (several records per read, strand ties, bounds that cut the seed search) through `kma` without -1t1 and
    through examples/kmahip_map -chain: .res, .fsa and .frag.gz. Seeds above 2 (KMA_CHAIN_FUZZ_SEEDS) draw the shape of the database,
    put N's into the reads (behind base 20: see the early-N note in DESIGN 3.1b) and draw a maxFrag."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    rng = np.random.default_rng(700 + seed)
    if seed <= 2:
        names, seqs = synth.make_gene_db(50, 5, 300, 900, 0.05, seed=800 + seed)
        n, with_n, extra = 30000, False, []
    else:
        names, seqs = synth.make_gene_db(int(rng.integers(5, 80)), int(rng.integers(2, 12)), 300, 1500, float(rng.choice([0.01, 0.04, 0.08])), seed=800 + seed)
        n, with_n, extra = 12000, True, ([] if seed % 3 == 0 else ["-mf", str(int(rng.integers(50, 5000)))])
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = _chimeric_reads(seqs, n, rng, with_n=with_n)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads)
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"] + extra, check=True, stderr=subprocess.DEVNULL)
    res, fsa, frag = _run_both(tmp_path, prefix, fq, extra)
    assert res == open(tmp_path / "ref.res", "rb").read()
    assert res.count(b"\n") > (100 if seed <= 2 else 5)
    assert fsa == open(tmp_path / "ref.fsa", "rb").read()
    ref_frag = gzip.open(tmp_path / "ref.frag.gz").read()
    if frag != ref_frag:
        a, b = frag.split(b"\n"), ref_frag.split(b"\n")
        d = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
        assert False, (len(a), len(b), len(d), a[d[0]][-120:] if d else None, b[d[0]][-120:] if d else None)
    assert frag.count(b"\n") > (10000 if seed <= 2 else 3000)
    # the same through batches of 1 777 reads with text chunks of 30 kB (records of a read never straddle batches, chunks of -mf do),
    # and through the one-batch call (kmahip_run_chain); the one batch with stage 2 in chunks of 3 001 reads (the anchors of a chunk made
    # beside the chaining of the chunk before, on a second set of buffers), and with the chunks one after the other
    for env in ({"KMAHIP_MAP_BATCH": "1777", "KMAHIP_FRAG_CHUNK": "30000"}, {"KMAHIP_MAP_ONE_BATCH": "1"}, {"KMAHIP_MAP_ONE_BATCH": "1", "KMAHIP_CHAIN_CHUNK": "3001"},
                {"KMAHIP_MAP_ONE_BATCH": "1", "KMAHIP_CHAIN_CHUNK": "3001", "KMAHIP_CHAIN_OVERLAP": "0"}):
        assert _run_both(tmp_path, prefix, fq, extra, env) == [res, fsa, frag], env



def test_default_mode_session_with_batches_that_hold_no_record_and_with_no_read_at_all(tmp_path):
    """the batched default mode (kmahip_session_set_chain) when a whole batch maps nowhere, when the input ends in such a batch, and
    when the input is empty: the files of the one-batch call"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    rng = np.random.default_rng(5)
    names, seqs = synth.make_gene_db(20, 4, 300, 800, 0.05, seed=91)
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    good = _chimeric_reads(seqs, 1500, rng)
    junk = [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(600)]
    for label, reads in (("middle", good[:500] + junk + good[500:]), ("end", good + junk), ("only", junk), ("empty", [])):
        fq = str(tmp_path / (label + ".fq"))
        synth.write_fastq(fq, reads)
        a = _run_both(tmp_path, prefix, fq, (), {"KMAHIP_MAP_BATCH": "500"})
        b = _run_both(tmp_path, prefix, fq, (), {"KMAHIP_MAP_ONE_BATCH": "1"})
        assert a == b, label
        if label in ("middle", "end"):
            subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            assert a[0] == open(tmp_path / "ref.res", "rb").read() and a[2] == gzip.open(tmp_path / "ref.frag.gz").read(), label
            assert a[2].count(b"\n") > 500


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("kind", ["genome", "genes"])
def test_long_read_route_equals_lane_kernel_and_oracle(tmp_path, kind):
    """stage 2 of long reads: chain_long_anchor_kernel (a wavefront per read and strand) + chain_long_tail_kernel against chain_kernel
    (KMAHIP_CHAIN=slow: a lane per read, anchors included) and the oracle -- lengths on either side of the fast route's limit and of
    the kernel's passes of 512 k-mer starts, reads shorter than k, unrelated reads, exact copies (one anchor spanning every pass), reads
    glued from pieces of both strands, exhaustive mode, and the pool cut into many chunks"""
    rng = np.random.default_rng(11)
    if kind == "genome":
        genome = rng.integers(0, 4, 300_000, dtype=np.uint8)
        names, seqs = ["g"], [genome]
    else:
        names, seqs = synth.make_gene_db(40, 6, 800, 3000, 0.03, seed=77)
        genome = np.concatenate(seqs)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads = []
    for L in (15, 16, 17, 303, 304, 305, 526, 527, 528, 529, 1038, 1039, 1040, 1041, 2000, 5000, 9000, 20000):
        for err in (0.0, 0.1):
            src = genome if kind == "genome" else seqs[int(rng.integers(0, len(seqs)))]
            if len(src) < L:
                src = genome
            st = int(rng.integers(0, len(src) - L + 1))
            w = src[st:st + L].copy()
            m = rng.random(L) < err
            w[m] = (w[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            reads.append(synth.revcomp_codes(w) if rng.random() < 0.5 else w)
    reads += synth.make_long_reads(genome, 120, read_len=6000, seed=3) + synth.make_long_reads(genome, 40, read_len=700, seed=4)
    reads += [rng.integers(0, 4, 4000, dtype=np.uint8) for _ in range(6)]
    for _ in range(30):          # pieces of both strands glued together: several chains per read
        parts = []
        for _ in range(int(rng.integers(2, 7))):
            src = seqs[int(rng.integers(0, len(seqs)))]
            L = int(rng.integers(200, min(2500, len(src))))
            st = int(rng.integers(0, len(src) - L + 1))
            w = src[st:st + L].copy()
            parts.append(synth.revcomp_codes(w) if rng.random() < 0.5 else w)
        reads.append(np.concatenate(parts))
    b = formats.pack_ragged(reads)
    odb = oracle.OracleDB(prefix)
    db = binding.KmaHipDB(prefix)
    try:
        for exhaustive in (0, 1):
            want = [(i, rf, er, qs, qe, tuple(int(t) for t in T)) for i, recs in enumerate(odb.scan_chain(b, exhaustive=exhaustive)) for rf, er, qs, qe, T in recs]
            slow = _with_env({"KMAHIP_CHAIN": "slow"}, lambda: _records(db.scan_chain(b, exhaustive=exhaustive)))
            off = _with_env({"KMAHIP_CHAIN_LONG": "0"}, lambda: _records(db.scan_chain(b, exhaustive=exhaustive)))
            got = _records(db.scan_chain(b, exhaustive=exhaustive))
            cut = _with_env({"KMAHIP_CHAIN_LONG_POOL_MB": "1"}, lambda: _records(db.scan_chain(b, exhaustive=exhaustive)))
            assert got == slow and off == slow and cut == slow, exhaustive
            assert got == want, exhaustive
            assert len(got) > 150
    finally:
        db.close()
