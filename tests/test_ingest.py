"""Stage-1 ingest (kmahip_ingest_*, host code in the C-ABI library) against the S1 streams the compiled reference wrote for
the same inputs (tests/golden/ingest, made by tests/golden/make_golden_ingest.py) and against the S1 taps of the mapping
fixtures. No GPU involved: FASTQ / FASTA parsing, phred-scale guess, quality trimming, length gate, 2-bit packing."""
import gzip
import os

import numpy as np
import pytest

from kma_amd import binding, formats, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ING = os.path.join(GOLD, "ingest")
SETTINGS = {
    "default": {},
    "mp25": dict(min_phred=25),
    "eq15": dict(min_q=15),
    "eq25ml40": dict(min_q=25, min_len=40),
    "mi12": dict(hardmask_q=12),
    "ml60xl130": dict(min_len=60, max_len=130),
}
CASES = {
    "p33": ("p33.fq", None, 33),
    "dos": ("dos.fq", None, 33),
    "p64": ("p64.fq", None, 64),
    "wrap": ("wrap.fa", None, None),
    "pe": ("m1.fq", "m2.fq", 33),
    "p33gz": ("p33gz.fq.gz", None, 33),
    "int": ("ilv.fq", "", 33),          # ("" = interleaved: the records of one file two at a time, run_input_INT)
    "intodd": ("ilvodd.fq", "", 33),
}


def _compare(batch, names, pair, s1):
    assert batch.n == len(s1)
    for i, r in enumerate(s1):
        L = int(batch.length[i])
        assert L == r["seqlen"], (i, L, r["seqlen"])
        w = batch.seq[batch.seq_off[i]:batch.seq_off[i] + ((L + 31) >> 5)]
        assert np.array_equal(w, r["seq"]), i
        assert batch.seq[batch.seq_off[i] + ((L + 31) >> 5)] == 0                  # the pad word
        assert np.array_equal(batch.N[batch.N_off[i]:batch.N_off[i + 1]], r["N"]), i
        assert names[i] + b"\0" == r["hdr"], (i, names[i], r["hdr"])
        assert (pair[i] == 1) == r["pair"], i
        if pair[i] == 1:
            assert pair[i + 1] == 2


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("setting", sorted(SETTINGS))
def test_ingest_matches_reference_s1(case, setting):
    f1, f2, phred = CASES[case]
    s1 = formats.parse_s1(gzip.open(os.path.join(ING, f"{case}.{setting}.s1.gz")).read())
    with binding.Ingest(os.path.join(ING, f1), os.path.join(ING, f2) if f2 else None, interleaved=f2 == "", **SETTINGS[setting]) as ing:
        if phred is not None:
            assert ing.phred_scale == phred
        got = ing.next(1 << 30)
        assert got is not None
        _compare(*got, s1)
        assert ing.next(10) is None
        read, kept = ing.counts()
        assert kept == sum(1 for i in range(got[0].n) if got[2][i] != 2)


def test_ingest_batches_are_a_partition():
    """small batches concatenate to the one-shot result (records never split: a pair stays in one batch)"""
    p1, p2 = os.path.join(ING, "m1.fq"), os.path.join(ING, "m2.fq")
    with binding.Ingest(p1, p2) as a:
        whole = a.next(1 << 30)
    parts = []
    with binding.Ingest(p1, p2) as b:
        while True:
            g = b.next(37)
            if g is None:
                break
            assert g[2][-1] != 1
            parts.append(g)
    assert sum(p[0].n for p in parts) == whole[0].n
    assert np.array_equal(np.concatenate([p[0].length for p in parts]), whole[0].length)
    assert [n for p in parts for n in p[1]] == whole[1]
    o = 0
    for p in parts:
        for i in range(p[0].n):
            L = int(p[0].length[i])
            a0, b0 = p[0].seq_off[i], whole[0].seq_off[o]
            assert np.array_equal(p[0].seq[a0:a0 + ((L + 31) >> 5)], whole[0].seq[b0:b0 + ((L + 31) >> 5)])
            o += 1


def test_ingest_batches_closed_by_their_bases(tmp_path, monkeypatch):
    """kmahip_ingest_set_batch_bases: a batch closes once it holds about that many bases (checked between the stretches of input the
    reader cuts into records); the batches still concatenate to the one-shot result"""
    monkeypatch.setenv("KMAHIP_INGEST_REGION", "256")           # small stretches: the bound is met often
    rng = np.random.default_rng(3)
    reads = [rng.integers(0, 4, int(rng.integers(200, 3000)), dtype=np.uint8) for _ in range(400)]
    fq = str(tmp_path / "long.fq")
    synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    with binding.Ingest(fq) as a:
        whole = a.next(1 << 30)
    parts = []
    with binding.Ingest(fq) as b:
        b.set_batch_bases(50_000)
        while True:
            g = b.next(1 << 30)
            if g is None:
                break
            parts.append(g)
    assert len(parts) > 5
    assert max(int(p[0].length.sum()) for p in parts) < 50_000 + 64 * 32 * 256         # (over by one stretch of input at most)
    assert np.array_equal(np.concatenate([p[0].length for p in parts]), whole[0].length)
    assert [n for p in parts for n in p[1]] == whole[1]
    with binding.Ingest(fq) as c:
        with pytest.raises(binding.KmaHipError):
            c.set_batch_bases(-1)


@pytest.mark.parametrize("name,files", [("se", ("reads.fq.gz", None)), ("long", ("reads.fq.gz", None)), ("pe", ("r1.fq.gz", "r2.fq.gz"))])
def test_ingest_matches_mapping_fixture_s1(name, files):
    src = os.path.join(GOLD, name)
    s1 = formats.parse_s1(gzip.open(os.path.join(src, "s1.bin.gz")).read())
    with binding.Ingest(os.path.join(src, files[0]), os.path.join(src, files[1]) if files[1] else None) as ing:
        _compare(*ing.next(1 << 30), s1)


def test_ingest_errors():
    with pytest.raises(binding.KmaHipError):
        binding.Ingest(os.path.join(ING, "does_not_exist.fq"))
    with pytest.raises(binding.KmaHipError):
        binding.Ingest(os.path.join(ING, "p33.fq"), os.path.join(ING, "wrap.fa"))       # different formats


def test_ingest_differential_against_reference_binary(tmp_path):
    """Seeded random FASTQ files (lengths 1 ... 400, qualities with bad ends / stretches, N / IUPAC / lower case, phred 33 and 64)
    and random trimming settings, through kmahip_ingest_* and through the compiled reference
    (`kma ... -s1`): the S1 records must agree. Skipped where oracle/_ref/kma has not been built. (Files whose last record has
    no final newline are left out: there the reference reads past the end of its buffer, seqparse.c:380-392; kmahip_ingest keeps the record.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kma = os.path.join(root, "oracle", "_ref", "kma")
    if not os.path.exists(kma):
        pytest.skip("oracle/_ref/kma not built")
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    import make_golden_ingest as mk
    from kma_amd import synth
    names, seqs = synth.make_gene_db(2, 2, 200, 300, 0.02, seed=1)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    rng = np.random.default_rng(2024)
    for case in range(30):
        fq = str(tmp_path / f"c{case}.fq")
        base = 64 if case % 5 == 4 else 33
        mk.write_fq(fq, int(rng.integers(1, 60)), 1000 + case, base=base, lens=(1, int(rng.integers(20, 400))))
        kw = dict(min_phred=int(rng.integers(0, 35)), min_q=int(rng.choice([0, 0, 10, 20, 28])), hardmask_q=int(rng.choice([0, 0, 5, 15])),
                  min_len=int(rng.integers(1, 80)), max_len=int(rng.choice([2**31 - 1, 150, 300])))
        cmd = [kma, "-i", fq, "-o", str(tmp_path / "o"), "-t_db", prefix, "-1t1", "-t", "1", "-s1", "-mp", str(kw["min_phred"]), "-eq", str(kw["min_q"]),
               "-mi", str(kw["hardmask_q"]), "-ml", str(kw["min_len"]), "-xl", str(kw["max_len"])]
        s1 = formats.parse_s1(subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout)
        with binding.Ingest(fq, None, **kw) as ing:
            got = ing.next(1 << 30)
        try:
            if got is None:
                assert len(s1) == 0
            else:
                _compare(*got, s1)
        except AssertionError as e:
            raise AssertionError(f"case {case} {kw}: {str(e)[:300]}")
        # the same file as interleaved input (`-int`: its records two at a time; an odd last one is filed singly)
        cmd[1] = "-int"
        s1 = formats.parse_s1(subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout)
        with binding.Ingest(fq, None, interleaved=True, **kw) as ing:
            got = ing.next(1 << 30)
        try:
            if got is None:
                assert len(s1) == 0
            else:
                _compare(*got, s1)
        except AssertionError as e:
            raise AssertionError(f"case {case} -int {kw}: {str(e)[:300]}")


def test_ingest_reports_malformed_input_after_the_good_records(tmp_path):
    fq = tmp_path / "bad.fq"
    fq.write_bytes(b"@a\nACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIII\nthis is not a record\n@b\nACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIII\n")
    with binding.Ingest(str(fq)) as ing:
        got = ing.next(100)
        assert got is not None and got[0].n == 1 and got[1] == [b"a"]
        with pytest.raises(binding.KmaHipError):
            ing.next(100)
        assert ing.next(100) is None


def _all(path1, path2=None, step=1 << 30, **kw):
    """every batch of a file concatenated: (lengths, names, pair flags, packed words per read); path2 == "": interleaved"""
    out = ([], [], [], [])
    with binding.Ingest(path1, path2 or None, interleaved=path2 == "", **kw) as ing:
        err = None
        while True:
            try:
                g = ing.next(step)
            except binding.KmaHipError as e:
                err = str(e)[:32]
                continue
            if g is None:
                break
            b, names, pair = g
            out[0].extend(int(x) for x in b.length)
            out[1].extend(names)
            out[2].extend(int(x) for x in pair)
            for i in range(b.n):
                L = int(b.length[i])
                out[3].append((bytes(b.seq[b.seq_off[i]:b.seq_off[i] + ((L + 31) >> 5)]), tuple(int(x) for x in b.N[b.N_off[i]:b.N_off[i + 1]])))
    return out, err


@pytest.fixture
def tiny_chunks(monkeypatch):
    """chunks of a few hundred bytes and regions of a few dozen: small files then run through every path of the chunked reader
    (records cut by chunk and region boundaries, records longer than a chunk, regions without a record start, wrong guesses)"""
    def set_(chunk, region, threads=4):
        monkeypatch.setenv("KMAHIP_INGEST_CHUNK", str(chunk))
        monkeypatch.setenv("KMAHIP_INGEST_REGION", str(region))
        monkeypatch.setenv("KMAHIP_INGEST_THREADS", str(threads))
    yield set_


@pytest.mark.parametrize("chunk,region", [(64, 16), (300, 40), (1000, 100), (4096, 64), (100000, 200)])
def test_ingest_chunked_reader_equals_one_pass(tmp_path, tiny_chunks, chunk, region):
    """the same files read whole (default sizes: one region each) and in tiny chunks / regions, plain and gzip-compressed"""
    import shutil
    import subprocess
    files = [("p33.fq", None), ("dos.fq", None), ("p64.fq", None), ("m1.fq", "m2.fq"), ("p33gz.fq.gz", None), ("ilv.fq", ""), ("ilvodd.fq", "")]
    # a file with hostile lines: qualities that begin with '@' and '+', a header that is only '@', long and empty reads
    rng = np.random.default_rng(5)
    rec = []
    for i in range(300):
        L = int(rng.choice([0, 1, 5, 30, 150, 700, 3000]))
        seq = "".join(rng.choice(list("ACGTN"), L))
        q = "".join(rng.choice(list("@+I5#"), L))
        rec.append(f"@{'r%d some text' % i if i % 7 else ''}\n{seq}\n+{'x' * int(rng.integers(0, 3))}\n{q}\n")
    hostile = tmp_path / "hostile.fq"
    hostile.write_text("".join(rec))
    nonl = tmp_path / "nonl.fq"
    nonl.write_text("".join(rec)[:-1])                     # last record without its final newline
    trunc = tmp_path / "trunc.fq"
    trunc.write_text("".join(rec)[:-40])                   # last record cut short
    files += [(str(hostile), None), (str(nonl), None), (str(trunc), None)]
    for f in (hostile, nonl, trunc):
        shutil.copy(f, str(f) + ".copy")
        subprocess.run(["gzip", "-1", str(f) + ".copy"], check=True)
        files.append((str(f) + ".copy.gz", None))
    for f1, f2 in files:
        p1 = f1 if os.path.isabs(f1) else os.path.join(ING, f1)
        p2 = os.path.join(ING, f2) if f2 else f2
        want = _all(p1, p2, min_phred=0, min_len=0)
        tiny_chunks(chunk, region)
        assert _all(p1, p2, min_phred=0, min_len=0) == want, (f1, "one batch")
        assert _all(p1, p2, step=7, min_phred=0, min_len=0) == want, (f1, "batches of 7")
        for k in ("KMAHIP_INGEST_CHUNK", "KMAHIP_INGEST_REGION", "KMAHIP_INGEST_THREADS"):
            os.environ.pop(k, None)


def test_ingest_interleaved_gzip_members_that_cut_a_couple(tmp_path, tiny_chunks):
    """the interleaved reader on a .gz of one member and on one of several members whose borders fall between the mates of a couple
    (an odd number of records in front of each border): the same batches as the plain file, whole and in tiny chunks"""
    raw = open(os.path.join(ING, "ilvodd.fq"), "rb").read()
    recs = raw.split(b"\n@q")
    recs = [recs[0] + b"\n"] + [b"@q" + r + b"\n" for r in recs[1:-1]] + [b"@q" + recs[-1]]
    assert b"".join(recs) == raw and len(recs) == 151
    (tmp_path / "one.fq.gz").write_bytes(gzip.compress(raw, 1))
    cuts = [0, 7, 40, 41, 96, 151]                                    # members of 7, 33, 1, 55 and 55 records
    (tmp_path / "many.fq.gz").write_bytes(b"".join(gzip.compress(b"".join(recs[a:b]), 1) for a, b in zip(cuts, cuts[1:])))
    want = _all(os.path.join(ING, "ilvodd.fq"), "", min_phred=0, min_len=1)
    assert sum(1 for x in want[0][2] if x == 1) > 50 and want[0][2][-1] == 0      # couples, and the odd last record filed singly
    for f in ("one.fq.gz", "many.fq.gz"):
        assert _all(str(tmp_path / f), "", min_phred=0, min_len=1) == want, f
        tiny_chunks(300, 40)
        assert _all(str(tmp_path / f), "", step=5, min_phred=0, min_len=1) == want, (f, "tiny chunks")
        for k in ("KMAHIP_INGEST_CHUNK", "KMAHIP_INGEST_REGION", "KMAHIP_INGEST_THREADS"):
            os.environ.pop(k, None)


def test_ingest_chunked_reader_malformed_and_short_mate(tmp_path, tiny_chunks):
    good = b"".join(b"@r%d\nACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIII\n" % i for i in range(200))
    bad = tmp_path / "bad.fq"
    bad.write_bytes(good + b"this is not a record\n" + good)
    short = tmp_path / "short.fq"
    short.write_bytes(good[:len(good) // 3 - 5])
    full = tmp_path / "full.fq"
    full.write_bytes(good)
    want_bad, want_pe = _all(str(bad)), _all(str(full), str(short))
    assert len(want_bad[0][0]) == 200 and want_bad[1] is not None
    assert len(want_pe[0][0]) > 200
    tiny_chunks(200, 30)
    assert _all(str(bad)) == want_bad
    assert _all(str(bad), step=11) == want_bad
    assert _all(str(full), str(short)) == want_pe
    assert _all(str(short), str(full), step=13) == _all(str(short), str(full))


def test_ingest_reports_a_corrupt_gzip_stream(tmp_path):
    """the records before the damage are delivered, then one call fails with an I/O error (not a silent end of input)"""
    import gzip as _gz
    rng = np.random.default_rng(3)
    lut = np.frombuffer(b"ACGT", np.uint8)
    good = b"".join(b"@r%d\n" % i + lut[rng.integers(0, 4, 150)].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(20000))
    z = bytearray(_gz.compress(good, 1))
    z[len(z) // 2:len(z) // 2 + 64] = bytes(64)           # damage in the middle of the deflate stream
    p = tmp_path / "bad.fq.gz"
    p.write_bytes(bytes(z))
    with binding.Ingest(str(p)) as ing:
        n, err = 0, None
        for _ in range(10):
            try:
                g = ing.next(1 << 30)
            except binding.KmaHipError as e:
                err = str(e)
                break
            if g is None:
                break
            n += g[0].n
        assert err is not None and 0 < n < 20000, (n, err)


def _member(data, level=1, fname=None):
    """one gzip member, hand-made so that the header can carry a file name"""
    import struct
    import zlib
    z = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = z.compress(data) + z.flush()
    head = b"\x1f\x8b\x08" + (b"\x08" if fname is not None else b"\x00") + b"\x00\x00\x00\x00" + (b"\x04" if level == 1 else b"\x00") + b"\x03"
    if fname is not None:
        head += fname + b"\x00"
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def _fastq(n, seed, L=150):
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGTN", np.uint8)
    return b"".join(b"@r%d\n" % i + lut[rng.integers(0, 5, int(rng.integers(20, L)))].tobytes() + b"\n+\n" for i in range(0)) + \
        b"".join(b"@r%d x\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in
                 ((i, lut[rng.integers(0, 5, int(rng.integers(20, L)))].tobytes()) for i in range(n)))


@pytest.mark.parametrize("chunk,limit", [(None, None), (500, 300), (4096, 1)])
def test_ingest_gzip_members_in_parallel(tmp_path, monkeypatch, chunk, limit):
    """a .gz of several members (cut anywhere, also inside records and lines) read member by member on several threads gives what the
    plain file gives: two big members, hundreds of small ones (bgzip-like), an empty member, a member whose header holds bytes that
    look like a member start, bytes behind the last member that are no member; with tiny chunks and a tiny read-ahead limit too"""
    data = _fastq(3000, 11)
    plain = tmp_path / "x.fq"
    plain.write_bytes(data)
    want = _all(str(plain), min_phred=0, min_len=0)
    assert len(want[0][0]) == 3000 and want[1] is None
    rng = np.random.default_rng(12)
    cases = {}
    cut = len(data) // 2 + 7
    cases["two"] = _member(data[:cut]) + _member(data[cut:], level=6)
    cuts = [0] + sorted(int(x) for x in rng.choice(len(data), 400, replace=False)) + [len(data)]
    cases["many"] = b"".join(_member(data[a:b]) for a, b in zip(cuts[:-1], cuts[1:]))
    cases["empty"] = _member(data[:1000]) + _member(b"") + _member(data[1000:])
    cases["decoy"] = _member(data[:5000], fname=b"a\x1f\x8b\x08\x01AAAA\x02\x03decoy-in-the-name") + _member(data[5000:], fname=b"\x1f\x8b\x08\x01BBBB\x04\x03")
    cases["garbage"] = _member(data[:cut]) + _member(data[cut:]) + b"\x00" * 100 + b"not a member"
    if chunk:
        monkeypatch.setenv("KMAHIP_INGEST_CHUNK", str(chunk))
        monkeypatch.setenv("KMAHIP_INGEST_REGION", "64")
        monkeypatch.setenv("KMAHIP_INGEST_GZ_LIMIT", str(limit))
    monkeypatch.setenv("KMAHIP_INGEST_THREADS", "5")
    for name, blob in cases.items():
        p = tmp_path / f"{name}.fq.gz"
        p.write_bytes(blob)
        if name != "garbage":
            assert gzip.open(p).read() == data, name          # (a well-formed multi-member file for any reader)
        assert _all(str(p), min_phred=0, min_len=0) == want, name
        assert _all(str(p), step=101, min_phred=0, min_len=0) == want, (name, "batches")
        monkeypatch.setenv("KMAHIP_INGEST_SERIAL_GZ", "1")    # the one-thread reader agrees
        assert _all(str(p), min_phred=0, min_len=0) == want, (name, "serial")
        monkeypatch.delenv("KMAHIP_INGEST_SERIAL_GZ")
    # pairs: both mate files multi-member, cut at different places
    d2 = _fastq(3000, 13)
    (tmp_path / "y.fq").write_bytes(d2)
    want_pe = _all(str(plain), str(tmp_path / "y.fq"), min_phred=0, min_len=0)
    (tmp_path / "y.fq.gz").write_bytes(_member(d2[:999]) + _member(d2[999:70000]) + _member(d2[70000:]))
    assert _all(str(tmp_path / "many.fq.gz"), str(tmp_path / "y.fq.gz"), min_phred=0, min_len=0) == want_pe


def test_ingest_gzip_members_damaged(tmp_path, monkeypatch):
    """a damaged member in the middle, and a file that ends inside its last member: the records before are delivered, then one call
    fails with an I/O error -- as the one-thread reader does"""
    monkeypatch.setenv("KMAHIP_INGEST_THREADS", "4")
    data = _fastq(6000, 21)
    third = len(data) // 3
    parts = [_member(data[:third]), bytearray(_member(data[third:2 * third])), _member(data[2 * third:])]
    mid = len(parts[1]) // 2
    parts[1][mid:mid + 64] = bytes(64)
    (tmp_path / "bad.fq.gz").write_bytes(parts[0] + bytes(parts[1]) + parts[2])
    whole = _member(data[:third]) + _member(data[third:])
    (tmp_path / "cut.fq.gz").write_bytes(whole[:-len(whole) // 4])
    first = data[:third].count(b"\n") // 4
    for name in ("bad", "cut"):
        got, err = _all(str(tmp_path / f"{name}.fq.gz"), min_phred=0, min_len=0)
        assert err is not None and first - 1 <= len(got[0]) < 6000, (name, len(got[0]), err)
        monkeypatch.setenv("KMAHIP_INGEST_SERIAL_GZ", "1")
        got1, err1 = _all(str(tmp_path / f"{name}.fq.gz"), min_phred=0, min_len=0)
        monkeypatch.delenv("KMAHIP_INGEST_SERIAL_GZ")
        k = min(len(got1[0]), len(got[0]))                 # (each drops a piece of its own size next to the damage: one is a prefix of the other)
        assert err1 is not None and got1[0][:k] == got[0][:k] and got1[1][:k] == got[1][:k], name
        assert "read error" in err, (name, err)


def test_ingest_gzip_member_the_scan_did_not_take_for_one(tmp_path, monkeypatch):
    """behind the last member the parallel reader recognised come bytes that begin like a gzip member but fail its stricter header test
    (an OS byte of 100; a member cut to under 18 bytes): gzread would inflate them or report the damage -- the reader must end with a
    read error, not drop them in silence as it does bytes that are no member at all"""
    monkeypatch.setenv("KMAHIP_INGEST_THREADS", "4")
    data = _fastq(3000, 22)
    cut = len(data) // 2
    odd = bytearray(_member(data[cut:]))
    odd[9] = 100                                           # OS byte: a valid member for zlib, not a candidate for the scan
    (tmp_path / "odd.fq.gz").write_bytes(_member(data[:cut // 2]) + _member(data[cut // 2:cut]) + bytes(odd))
    (tmp_path / "short.fq.gz").write_bytes(_member(data[:cut // 2]) + _member(data[cut // 2:cut]) + _member(data[cut:])[:12])
    for name in ("odd", "short"):
        got, err = _all(str(tmp_path / f"{name}.fq.gz"), min_phred=0, min_len=0)
        assert err is not None and "read error" in err, (name, err)
        assert len(got[0]) >= data[:cut].count(b"\n") // 4 - 1
