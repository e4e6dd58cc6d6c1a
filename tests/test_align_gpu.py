"""Stage-3a parity on the GPU: HIP path (through the C-ABI) vs the reference frag_raw taps and the CPU oracle."""
import numpy as np
import pytest

import golden_util
from kma_amd import formats, synth

pytestmark = pytest.mark.gpu


def _run(g):
    from kma_amd import binding
    db = binding.KmaHipDB(g["prefix"])
    try:
        return db.map_se(g["batch"])
    finally:
        db.close()


def test_align_matches_reference_frag_raw_tap(golden_se):
    (rc_flag, flag, T_off, T), h = _run(golden_se)
    golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2"], rc_flag, flag, T_off, T)
    n = golden_util.check_align_against_frag_raw(golden_se["s1"], golden_util.load_frag_raw("se"), T_off, h)
    assert n > 900


def test_align_long_reads_banded_matches_reference_frag_raw_tap(golden_long):
    (rc_flag, flag, T_off, T), h = _run(golden_long)
    golden_util.check_scan_against_s2(golden_long["s1"], golden_long["s2"], rc_flag, flag, T_off, T)
    n = golden_util.check_align_against_frag_raw(golden_long["s1"], golden_util.load_frag_raw("long"), T_off, h)
    assert n > 200


def _vs_oracle(prefix, batch):
    import oracle
    from kma_amd import binding
    db = binding.KmaHipDB(prefix)
    try:
        (rc_flag, flag, T_off, T), h = db.map_se(batch)
    finally:
        db.close()
    odb = oracle.OracleDB(prefix)
    e = odb.scan_se(batch)
    for a, b in zip(e, (rc_flag, flag, T_off, T)):
        assert np.array_equal(a, b)
    o = odb.align_se(batch, rc_flag, flag, T_off, T)
    assert np.array_equal(o["n_hits"], h["n_hits"])
    assert np.array_equal(o["best_score"], h["best_score"])
    assert np.array_equal(o["out_flag"], h["flag"])
    for i in np.nonzero(o["n_hits"] > 0)[0]:
        s, c = int(T_off[i]), int(o["n_hits"][i])
        for key in ("tmpl", "start", "end", "score"):
            assert np.array_equal(o[key][s:s + c], h[key][s:s + c]), (i, key)
    assert np.array_equal(o["alignment_scores"], h["alignment_scores"])
    assert np.array_equal(o["uniq_alignment_scores"], h["uniq_alignment_scores"])
    return o


def test_align_vs_oracle_5variant_db(tmp_path):
    names, seqs = synth.make_gene_db(n_families=40, variants=5, seed=17)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads, *_ = synth.make_reads(seqs, 6000, read_len=150, sub_rate=0.01, random_frac=0.03, n_rate=0.001, seed=9)
    o = _vs_oracle(prefix, formats.pack_fixed(reads))
    assert (o["n_hits"] > 0).sum() > 5000
    assert o["alignment_scores"].sum() > 0 and o["uniq_alignment_scores"].sum() > 0


def test_align_vs_oracle_indels_and_ragged(tmp_path):
    names, seqs = synth.make_gene_db(n_families=12, variants=4, len_lo=800, len_hi=2500, seed=23)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    cat = np.concatenate(seqs)
    reads = synth.make_long_reads(cat, 300, read_len=400, sub=0.03, dele=0.02, ins=0.02, seed=4)
    short, *_ = synth.make_reads(seqs, 500, read_len=75, sub_rate=0.03, seed=6)
    reads = reads + list(short)
    _vs_oracle(prefix, formats.pack_ragged(reads))


@pytest.mark.parametrize("route", ["general", "fast"])
def test_both_stage3a_kernels_equal_the_oracle(tmp_path, monkeypatch, route, golden_se):
    """align_fast_kernel (MEMs, chain and sums in registers; hands the other tasks on) + align_tasks_kernel over what it hands on,
    against align_tasks_kernel for every task (KMAHIP_ALIGN_FAST=0): the oracle and the reference taps must come out either way --
    reads with 1-3 % substitutions (several MEMs per task), N's (seeded by the general kernel), indels (DP problems for the queues),
    reads that hang over template ends, strand ties in an inverted repeat"""
    monkeypatch.setenv("KMAHIP_ALIGN_FAST", "0" if route == "general" else "1")
    names, seqs = synth.make_gene_db(n_families=30, variants=4, len_lo=300, len_hi=1200, seed=41)
    rng = np.random.default_rng(8)
    pal = seqs[3][:200]
    seqs.append(np.concatenate([pal, rng.integers(0, 4, 40, dtype=np.uint8), (3 - pal)[::-1]]))      # an inverted repeat: strand ties
    names.append("inverted")
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads = []
    for sub, n_rate in ((0.01, 0.0), (0.03, 0.002), (0.002, 0.0)):
        r, *_ = synth.make_reads(seqs, 2500, read_len=150, sub_rate=sub, random_frac=0.02, n_rate=n_rate, seed=int(sub * 1000) + 3)
        reads += list(r)
    cat = np.concatenate(seqs)
    reads += synth.make_long_reads(cat, 400, read_len=180, sub=0.01, dele=0.01, ins=0.01, seed=12)      # across template ends, with indels
    reads += [seqs[-1][a:a + 150].copy() for a in rng.integers(0, len(seqs[-1]) - 150, 60)]
    _vs_oracle(prefix, formats.pack_ragged(reads))
    (rc_flag, flag, T_off, T), h = _run(golden_se)
    assert golden_util.check_align_against_frag_raw(golden_se["s1"], golden_util.load_frag_raw("se"), T_off, h) > 900


def test_shard_invariance_and_determinism_large_batch(tmp_path):
    """Size-independent properties on a batch far larger than the oracle cases: the ConClave vectors of two
    read shards add up to the whole batch's, per-read results do not depend on batching, and a second run is
    bit-identical (atomics only feed exact integer sums)."""
    from kma_amd import binding
    names, seqs = synth.make_gene_db(n_families=200, variants=5, seed=5)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads, *_ = synth.make_reads(seqs, 300_000, read_len=150, sub_rate=0.005, random_frac=0.02, seed=21)
    whole = formats.pack_fixed(reads)
    half = len(reads) // 2 + 37
    db = binding.KmaHipDB(prefix)
    try:
        (rc1, fl1, To1, T1), h1 = db.map_se(whole)
        (rc2, fl2, To2, T2), h2 = db.map_se(whole)
        (_, _, ToA, TA), hA = db.map_se(formats.pack_fixed(reads[:half]))
        (_, _, ToB, TB), hB = db.map_se(formats.pack_fixed(reads[half:]))
    finally:
        db.close()
    for a, b in ((rc1, rc2), (fl1, fl2), (To1, To2), (T1, T2), (h1["n_hits"], h2["n_hits"]), (h1["best_score"], h2["best_score"]),
                 (h1["alignment_scores"], h2["alignment_scores"]), (h1["uniq_alignment_scores"], h2["uniq_alignment_scores"])):
        assert np.array_equal(a, b)
    assert np.array_equal(h1["alignment_scores"], hA["alignment_scores"] + hB["alignment_scores"])
    assert np.array_equal(h1["uniq_alignment_scores"], hA["uniq_alignment_scores"] + hB["uniq_alignment_scores"])
    assert np.array_equal(h1["n_hits"], np.concatenate([hA["n_hits"], hB["n_hits"]]))
    assert np.array_equal(h1["best_score"], np.concatenate([hA["best_score"], hB["best_score"]]))
    assert np.array_equal(T1, np.concatenate([TA, TB]))
    # every kept hit carries the read's best score or the best score/length ratio; mapped fraction is sane
    assert (h1["n_hits"] > 0).mean() > 0.95
    assert int(h1["alignment_scores"].sum()) >= int(h1["best_score"].astype(np.int64).sum())


def test_align_vs_oracle_long_noisy_reads(tmp_path):
    """5-8 kb ONT-like reads against one 120 kb sequence: multi-pass scan, hundreds of MEMs per task, banded NW
    with wide bands, DP rows in HBM scratch."""
    rng = np.random.default_rng(8)
    genome = rng.integers(0, 4, 120_000, dtype=np.uint8)
    prefix = str(tmp_path / "g")
    formats.write_index(prefix, ["genome"], [genome])
    reads = []
    for L in (5000, 6500, 8000):
        reads += synth.make_long_reads(genome, 6, read_len=L, sub=0.03, dele=0.02, ins=0.02, seed=int(rng.integers(1 << 30)))
    _vs_oracle(prefix, formats.pack_ragged(reads))


def test_align_vs_oracle_partially_matching_reads(tmp_path):
    """Reads that match a template only in part (a gene segment followed or preceded by 60 ... 230 foreign bases, both
    strands, some with a second segment further on): the unaligned ends are DP problems of 64 ... 255 query columns, solved
    by the extra-wide cooperative path (nw_coop_x), up to the ones only a single lane or the banded DP can take."""
    names, seqs = synth.make_gene_db(n_families=12, variants=3, len_lo=700, len_hi=1100, seed=23)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    rng = np.random.default_rng(99)
    reads = []
    for i in range(1500):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(250, 450))
        a = int(rng.integers(0, len(s) - L))
        core = s[a:a + L].copy()
        junk = rng.integers(0, 4, int(rng.integers(64, 150)), dtype=np.uint8)
        kind = i % 4
        if kind == 0:
            r = np.concatenate([core, junk])
        elif kind == 1:
            r = np.concatenate([junk, core])
        elif kind == 2:
            r = np.concatenate([junk[:len(junk) // 2], core, junk[len(junk) // 2:]])
        else:                      # two segments of the same gene around the junk: a wide gap between two MEM chains
            b = min(len(s) - 30, a + L + int(rng.integers(0, 80)))
            r = np.concatenate([core, junk[:int(rng.integers(64, len(junk) + 1))], s[b:b + 30]])
        if rng.random() < 0.01 * 30:
            p = int(rng.integers(0, len(r)))
            r[p] = (r[p] + 1) & 3
        if rng.random() < 0.5:
            r = synth.revcomp_codes(r)
        reads.append(np.ascontiguousarray(r.astype(np.uint8)))
    o = _vs_oracle(prefix, formats.pack_ragged(reads))
    assert (o["n_hits"] > 0).sum() > 300, int((o["n_hits"] > 0).sum())


def test_align_vs_oracle_long_unaligned_ends_banded(tmp_path):
    """Ends of 129 ... 200 foreign bases next to 500 ... 750 matching ones: both sides of the end problem exceed the band
    (|dt - dq| + 64), so it is NW_band_score's -- solved on the cooperative sweep in query coordinates (nw_coop_x, band > 0),
    incl. free-end modes at template starts / ends. The reads keep enough score to be hits, so the DP results are observable."""
    names, seqs = synth.make_gene_db(n_families=10, variants=2, len_lo=1400, len_hi=2000, seed=31)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    rng = np.random.default_rng(7)
    reads = []
    for i in range(1200):
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(500, 750))
        a = int(rng.integers(0, len(s) - L)) if i % 5 else (0 if i % 10 else len(s) - L)     # every fifth at a template end
        core = s[a:a + L].copy()
        junk = rng.integers(0, 4, int(rng.integers(129, 201)), dtype=np.uint8)
        r = np.concatenate([core, junk]) if i % 2 else np.concatenate([junk, core])
        for _ in range(int(rng.integers(0, 3))):
            p = int(rng.integers(0, len(r)))
            r[p] = (r[p] + 1) & 3
        if rng.random() < 0.5:
            r = synth.revcomp_codes(r)
        reads.append(np.ascontiguousarray(r.astype(np.uint8)))
    o = _vs_oracle(prefix, formats.pack_ragged(reads))
    assert (o["n_hits"] > 0).sum() > 600, int((o["n_hits"] > 0).sum())
