"""kmahip_index_build against the reference's `kma index` on the same FASTA: the same k-mer -> template-list mapping, the same
.length.b / .seq.b / .name, and -- what an index is for -- the reference binary and the HIP path map a read set against either index with
identical results."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from kma_amd import binding, formats, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")


def _fasta_with_oddities(path, names, seqs, rng):
    """templates with leading / trailing / inner N's, IUPAC codes, lower case, wrapped lines, one too short, trailing blanks in a header"""
    lut = np.frombuffer(b"ACGT", np.uint8)
    with open(path, "wb") as f:
        for i, (n, s) in enumerate(zip(names, seqs)):
            t = bytearray(lut[s].tobytes())
            if i % 7 == 1:
                t = bytearray(b"NNN") + t + bytearray(b"NN")
            if i % 7 == 2:
                for p in rng.integers(40, len(t) - 40, 3):
                    t[int(p)] = ord("N")
            if i % 7 == 3:
                for p, c in zip(rng.integers(40, len(t) - 40, 4), b"RYKM"):
                    t[int(p)] = c
            if i % 7 == 4:
                t = bytearray(bytes(t).lower())
            f.write(b">" + n.encode() + (b"  \n" if i % 5 == 0 else b"\n"))
            w = 60 if i % 3 else len(t)
            for a in range(0, len(t), w):
                f.write(bytes(t[a:a + w]) + b"\n")
        f.write(b">too_short\nACGTACGT\n")


@pytest.mark.parametrize("k", [16, 12])
def test_index_build_equals_kma_index(tmp_path, k):
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    rng = np.random.default_rng(k)
    names, seqs = synth.make_gene_db(40, 5, 400, 900, 0.04, seed=500 + k)
    fa = str(tmp_path / "t.fsa")
    _fasta_with_oddities(fa, names, seqs, rng)
    ref, got = str(tmp_path / "ref"), str(tmp_path / "got")
    subprocess.run([KMA, "index", "-i", fa, "-o", ref, "-k", str(k)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    binding.index_build(fa, got, k=k)
    for ext in (".length.b", ".name"):
        assert open(got + ext, "rb").read() == open(ref + ext, "rb").read(), ext
    # .seq.b: (len >> 5) + 1 words per template; when len is a multiple of 32 the last one is whatever the reference's buffer held
    # before (updateindex.c:172 writes one word more than compDNAref filled), so only the words that carry bases are compared
    lens = formats.read_lengths(ref)[1:]
    wa, wb = np.fromfile(got + ".seq.b", np.uint64), np.fromfile(ref + ".seq.b", np.uint64)
    assert len(wa) == len(wb) == int(((lens.astype(np.int64) >> 5) + 1).sum())
    o = 0
    for L in lens.tolist():
        w = (L + 31) >> 5
        assert np.array_equal(wa[o:o + w], wb[o:o + w])
        o += (L >> 5) + 1
    a, b = formats.read_comp_b(got + ".comp.b"), formats.read_comp_b(ref + ".comp.b")
    assert (a.DB_size, a.mlen, a.kmersize, a.flag, a.n, a.prefix_len) == (b.DB_size, b.mlen, b.kmersize, b.flag, b.n, b.prefix_len)
    assert formats.comp_db_mapping(a) == formats.comp_db_mapping(b)
    assert a.v_index == b.v_index                                    # equal lists are stored once in both
    # mapping against either index: the reference binary, and the HIP path
    reads, *_ = synth.make_reads(seqs, 20000, read_len=120, sub_rate=0.01, random_frac=0.03, seed=9)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    outs = {}
    for tag, idx in (("ref", ref), ("got", got)):
        subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / f"kma_{tag}"), "-t_db", idx, "-1t1", "-t", "1"], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", idx, "-o", str(tmp_path / f"hip_{tag}"), "-1t1"], check=True,
                       stderr=subprocess.DEVNULL)
        outs[tag] = [open(tmp_path / f"{p}_{tag}.res", "rb").read() for p in ("kma", "hip")] + \
                    [gzip.open(tmp_path / f"{p}_{tag}.frag.gz").read() for p in ("kma", "hip")]
    assert outs["ref"][0] == outs["got"][0] == outs["ref"][1] == outs["got"][1] and outs["ref"][0].count(b"\n") > 30
    assert outs["ref"][2] == outs["got"][2] == outs["ref"][3] == outs["got"][3]


def test_index_build_errors(tmp_path):
    with pytest.raises(binding.KmaHipError):
        binding.index_build(str(tmp_path / "missing.fsa"), str(tmp_path / "x"))
    p = tmp_path / "short.fsa"
    p.write_text(">a\nACGT\n")
    with pytest.raises(binding.KmaHipError):
        binding.index_build(str(p), str(tmp_path / "x"))
    with pytest.raises(binding.KmaHipError):
        binding.index_build(str(p), str(tmp_path / "x"), k=20)


def test_index_host_program_takes_a_batch_file(tmp_path):
    """`-batch list.txt` (index.c:351-401: a file of input paths, one a line) against `kma index -batch` and against the same files given with -i"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    names, seqs = synth.make_gene_db(30, 4, 400, 900, 0.04, seed=77)
    paths = []
    for part in range(3):
        p = str(tmp_path / f"part{part}.fsa")
        synth.write_fasta(p, names[part::3], seqs[part::3])
        paths.append(p)
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    subprocess.run([KMA, "index", "-batch", str(tmp_path / "list.txt"), "-o", str(tmp_path / "ref")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_index"), "-batch", str(tmp_path / "list.txt"), "-o", str(tmp_path / "got")], check=True)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_index"), "-i"] + paths + ["-o", str(tmp_path / "got_i")], check=True)
    for ext in (".length.b", ".name"):
        assert open(str(tmp_path / "got") + ext, "rb").read() == open(str(tmp_path / "ref") + ext, "rb").read(), ext
    a, b, c = (formats.read_comp_b(str(tmp_path / t) + ".comp.b") for t in ("got", "ref", "got_i"))
    assert formats.comp_db_mapping(a) == formats.comp_db_mapping(b) and a.v_index == b.v_index
    for ext in (".comp.b", ".length.b", ".name", ".seq.b"):
        assert open(str(tmp_path / "got") + ext, "rb").read() == open(str(tmp_path / "got_i") + ext, "rb").read(), ext
    assert c.n == a.n


def test_index_host_program_and_speed(tmp_path):
    """examples/kmahip_index on the 5 k-gene database of the benchmark: loads, and is timed next to `kma index` (printed with -s)"""
    import time
    names, seqs = synth.make_gene_db(1000, 5, 600, 1500, 0.04, seed=12345)
    fa = str(tmp_path / "db5k.fsa")
    synth.write_fasta(fa, names, seqs)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    t0 = time.perf_counter()
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_index"), "-i", fa, "-o", str(tmp_path / "got")], check=True)
    t_hip = time.perf_counter() - t0
    t_ref = None
    if os.path.exists(KMA):
        t0 = time.perf_counter()
        subprocess.run([KMA, "index", "-i", fa, "-o", str(tmp_path / "ref")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t_ref = time.perf_counter() - t0
        a, b = formats.read_comp_b(str(tmp_path / "got.comp.b")), formats.read_comp_b(str(tmp_path / "ref.comp.b"))
        assert a.n == b.n and a.v_index == b.v_index
    print(f"kmahip_index {t_hip:.2f} s (whole process), kma index {t_ref if t_ref is None else round(t_ref, 2)} s")
    db = binding.KmaHipDB(str(tmp_path / "got"))
    assert int(db.info.DB_size) == 5001
    db.close()
