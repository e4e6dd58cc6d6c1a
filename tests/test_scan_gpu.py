"""Stage-2 parity on the GPU: HIP path (through the C-ABI) vs reference taps and vs the CPU oracle."""
import numpy as np
import pytest

import golden_util
from kma_amd import formats, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hipdb(golden_se):
    from kma_amd import binding
    db = binding.KmaHipDB(golden_se["prefix"])
    yield db
    db.close()


def test_scan_matches_reference_s2_tap(golden_se, hipdb):
    rc_flag, flag, T_off, T = hipdb.scan_se(golden_se["batch"])
    n = golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2"], rc_flag, flag, T_off, T)
    assert n > 800


def test_scan_exhaustive_matches_reference_s2_tap(golden_se, hipdb):
    rc_flag, flag, T_off, T = hipdb.scan_se(golden_se["batch"], exhaustive=1)
    golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2_ex"], rc_flag, flag, T_off, T)


def test_scan_empty_and_tiny_batches(golden_se, hipdb):
    empty = formats.pack_ragged([])
    rc_flag, flag, T_off, T = hipdb.scan_se(empty)
    assert len(rc_flag) == 0 and T_off.tolist() == [0] and len(T) == 0
    one = formats.pack_ragged([golden_se["reads"][0]])
    rc_flag, flag, T_off, T = hipdb.scan_se(one)
    assert T_off[1] == len(T)


def _compare_with_oracle(prefix, batch, hipdb, exhaustive=0):
    import oracle
    odb = oracle.OracleDB(prefix)
    e = odb.scan_se(batch, exhaustive=exhaustive)
    g = hipdb.scan_se(batch, exhaustive=exhaustive)
    assert np.array_equal(e[2], g[2]), "T_off differs"
    assert np.array_equal(e[0], g[0]), "rc_flag differs"
    assert np.array_equal(e[1], g[1]), "flag differs"
    assert np.array_equal(e[3], g[3]), "T differs"
    return e


@pytest.mark.parametrize("variants", [40, 90])
def test_scan_vs_oracle_redundant_db(tmp_path, variants):
    """Wide candidate sets: 40 variants per family do not fit the 16-slot candidate tables and go through the second tier
    (64 slots); 90 variants do not fit those either and force the HBM overflow path (one wavefront per item)."""
    from kma_amd import binding
    names, seqs = synth.make_gene_db(n_families=6, variants=variants, len_lo=500, len_hi=900, max_div=0.03, seed=99)
    prefix = str(tmp_path / "red")
    formats.write_index(prefix, names, seqs)
    reads, *_ = synth.make_reads(seqs, 3000, read_len=150, sub_rate=0.01, random_frac=0.03, n_rate=0.002, seed=5)
    batch = formats.pack_fixed(reads)
    db = binding.KmaHipDB(prefix)
    try:
        e = _compare_with_oracle(prefix, batch, db)
        counts = np.diff(e[2])
        assert counts.max() >= 2
    finally:
        db.close()


def test_scan_vs_oracle_ragged_lengths(tmp_path):
    from kma_amd import binding
    names, seqs = synth.make_gene_db(n_families=30, variants=5, seed=3)
    prefix = str(tmp_path / "rag")
    formats.write_index(prefix, names, seqs)
    rng = np.random.default_rng(11)
    reads = []
    for L in (15, 16, 17, 31, 32, 33, 64, 136, 150, 151, 152, 300, 600):
        r, *_ = synth.make_reads([s for s in seqs if len(s) >= L], 60, read_len=L, sub_rate=0.02,
                                 random_frac=0.05, n_rate=0.004, seed=int(rng.integers(1 << 30)))
        reads.extend(list(r))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    batch = formats.pack_ragged(reads)
    db = binding.KmaHipDB(prefix)
    try:
        _compare_with_oracle(prefix, batch, db)
        _compare_with_oracle(prefix, batch, db, exhaustive=1)
    finally:
        db.close()


def test_c_host_program_reproduces_reference_s2_stream(golden_se):
    """examples/kmahip_s2.c (plain C99 over the C-ABI): the reference's S1 stream in, the reference's S2 stream out."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    exe = os.path.join(root, "examples", "kmahip_s2")
    src = golden_se["dir"]
    s1 = gzip.open(os.path.join(src, "s1.bin.gz"), "rb").read()
    for extra, name in (([], "s2.bin.gz"), (["-ex_mode"], "s2_ex.bin.gz")):
        out = subprocess.run([exe, "-t_db", golden_se["prefix"]] + extra, input=s1, stdout=subprocess.PIPE, check=True).stdout
        assert out == gzip.open(os.path.join(src, name), "rb").read(), name


def test_prefilter_with_and_without_presence_bits(golden_se, monkeypatch):
    """Small databases get presence bits in front of the prefilter's probe-table gathers; the result may not depend on them."""
    from kma_amd import binding
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("KMAHIP_NO_KBITS", "1")
        db = binding.KmaHipDB(golden_se["prefix"])
        try:
            outs.append(db.scan_se(golden_se["batch"]))
        finally:
            db.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
