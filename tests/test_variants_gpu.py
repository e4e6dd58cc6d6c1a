"""Index variants of the same file format: k-mer sizes below 16 and databases wide enough for 32-bit value lists
(DB_size >= 65535, hashmapkma.c:340-348). HIP path vs the oracle on seeded inputs."""
import numpy as np
import pytest

import oracle
from kma_amd import formats, synth

pytestmark = pytest.mark.gpu


def _compare(prefix, reads, exhaustive=0):
    from kma_amd import binding
    b = formats.pack_ragged(reads)
    db = binding.KmaHipDB(prefix)
    try:
        (rc_flag, flag, T_off, T), h = db.map_se(b, exhaustive=exhaustive)
        cc = db.conclave_se(b.length, T_off, h)
    finally:
        db.close()
    odb = oracle.OracleDB(prefix)
    exp = odb.scan_se(b, exhaustive=exhaustive)
    for g, e, nm in zip((rc_flag, flag, T_off, T), exp, ("rc_flag", "flag", "T_off", "T")):
        assert np.array_equal(g, e), nm
    o = odb.align_se(b, *exp)
    keep = o["n_hits"] >= 0                      # the oracle does not restate strand ties
    assert np.array_equal(o["n_hits"][keep], h["n_hits"][keep])
    assert np.array_equal(o["best_score"][keep], h["best_score"][keep])
    if keep.all():
        for key in ("tmpl", "start", "end", "score"):
            assert np.array_equal(o[key][:len(T)], h[key][:len(T)]), key
        assert np.array_equal(o["alignment_scores"], h["alignment_scores"])
        tlen = formats.read_lengths(prefix)
        oc = oracle.conclave(o["n_hits"], o["best_score"], b.length, np.zeros(b.n, np.int32), exp[2][:-1], o["tmpl"], o["start"],
                             o["end"], o["alignment_scores"], o["uniq_alignment_scores"], tlen)
        assert np.array_equal(cc["tmpl"], oc["tmpl"]) and np.array_equal(cc["w_scores"], oc["w_scores"])
    return int((h["n_hits"] > 0).sum())


@pytest.mark.parametrize("k", [11, 12, 14])
def test_small_kmer_index(tmp_path, k):
    names, seqs = synth.make_gene_db(30, 4, 300, 700, 0.04, seed=100 + k)
    prefix = str(tmp_path / f"k{k}")
    formats.write_index(prefix, names, seqs, k=k)
    reads, *_ = synth.make_reads(seqs, 3000, read_len=100, sub_rate=0.02, random_frac=0.05, n_rate=0.002, seed=k)
    assert _compare(prefix, list(reads)) > 2000


def test_wide_database_u32_value_lists(tmp_path):
    # 66 000 templates -> DB_size >= 65535 -> the index stores u32 value lists
    names, seqs = synth.make_gene_db(16500, 4, 48, 72, 0.05, seed=9)
    prefix = str(tmp_path / "wide")
    formats.write_index(prefix, names, seqs)
    rng = np.random.default_rng(3)
    reads = []
    for _ in range(4000):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(30, len(s) + 1))
        st = int(rng.integers(0, len(s) - L + 1))
        r = s[st:st + L].copy()
        m = rng.random(L) < 0.01
        r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
        reads.append(synth.revcomp_codes(r) if rng.random() < 0.5 else r)
    assert _compare(prefix, reads) > 3000
