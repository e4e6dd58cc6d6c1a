"""The long-read traceback's lane-per-problem kernels (lt_lane_kernel / lt_lane_band_kernel, longtrace.hip) against the wave-per-problem
kernels they took the work from: the same reads through kmahip_align_trace_mt1 with the lane classes on, off, full-matrix only, banded
only, with the score table instead of the computed score, under a scoring scheme that is not three-valued, and under penalties so large
that 16 bits no longer hold the scores (then the lane classes must stand back by themselves). Figures, strands and alignment runs must be
identical in every arrangement; the arrangement with everything off is the one the reference-binary tests pinned in round 2."""
import os
import subprocess

import numpy as np
import pytest

from kma_amd import formats, synth

pytestmark = pytest.mark.gpu
KMA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "kma")


def _reads(genome, rng):
    """ONT-like reads of 300 .. 9000 bases in both orientations, some with N's, some with a long foreign stretch (wide problems), some
    with a long deletion or insertion (banded problems with a wide band)"""
    base = synth.make_long_reads(genome, 160, read_len=6000, seed=int(rng.integers(1 << 30)))
    out = []
    for i, r in enumerate(base):
        r = r.copy()
        L = int(rng.integers(300, len(r)))
        r = r[:L]
        if i % 5 == 0 and L > 200:                       # N's
            pos = rng.integers(0, L, 4)
            r[pos] = 4
        if i % 7 == 0 and L > 1500:                      # a foreign stretch of 100 .. 400 bases
            a = int(rng.integers(300, L - 600)); w = int(rng.integers(100, 400))
            r[a:a + w] = rng.integers(0, 4, w, dtype=np.uint8)
        if i % 11 == 0 and L > 2000:                     # a deletion of 20 .. 70 bases in the read
            a = int(rng.integers(500, L - 600)); w = int(rng.integers(20, 70))
            r = np.concatenate([r[:a], r[a + w:]])
        if i % 13 == 0 and L > 2000:                     # an insertion
            a = int(rng.integers(500, L - 600)); w = int(rng.integers(20, 70))
            r = np.concatenate([r[:a], rng.integers(0, 4, w, dtype=np.uint8), r[a:]])
        out.append(r)
    return out


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    tmp = tmp_path_factory.mktemp("lane")
    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 300_000, dtype=np.uint8)
    prefix = str(tmp / "g")
    synth.write_fasta(prefix + ".fsa", ["genome"], [genome])
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = _reads(genome, rng)
    return prefix, formats.pack_ragged(reads)


def _run(prefix, batch, env, tweak=None):
    from kma_amd import binding
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    db = binding.KmaHipDB(prefix)
    try:
        if tweak:
            tweak(db.params)
        (stats, off, nops, ops), rc = db.align_trace_mt1(batch, 1)
        st = db.longtrace_stats() if hasattr(db, "longtrace_stats") else None
        return stats.copy(), off.copy(), nops.copy(), ops.copy(), rc.copy(), st
    finally:
        db.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same(a, b, what):
    assert np.array_equal(a[4], b[4]), what + ": strands"
    assert np.array_equal(a[0], b[0]), what + ": figures " + str(np.argwhere(a[0] != b[0])[:5].tolist())
    assert np.array_equal(a[2], b[2]), what + ": run counts"
    for i in range(len(a[2])):
        ra, rb = a[3][a[1][i]:a[1][i] + a[2][i]], b[3][b[1][i]:b[1][i] + b[2][i]]
        assert np.array_equal(ra, rb), f"{what}: runs of read {i}"


def test_lane_classes_equal_the_wave_per_problem_kernels(setup):
    prefix, batch = setup
    ref = _run(prefix, batch, {"KMAHIP_LT_LANE": "0"})
    assert int((ref[0][:, 3] > 0).sum()) > 100          # most reads align
    for env in ({}, {"KMAHIP_LT_LCLS": "0x1ff"}, {"KMAHIP_LT_REG128": "0"}, {"KMAHIP_LT_PASS_READS": "23"}, {"KMAHIP_LT_PASS_READS": "23", "KMAHIP_LT_PIPE": "0"}, {"KMAHIP_LT_LANE": "f"}, {"KMAHIP_LT_LANE": "b"}, {"KMAHIP_LT_SCORE_TABLE": "1"}, {"KMAHIP_LT_SEED_WGS": "7"}, {"KMAHIP_LT_REGBAND": "0"}, {"KMAHIP_LT_REG": "0"}):
        _same(ref, _run(prefix, batch, env), str(env))


def test_lane_classes_under_a_scheme_that_is_not_three_valued(setup):
    prefix, batch = setup

    def tweak(p):
        # transitions cost less than transversions: the score of a pair comes out of the table
        for i in range(4):
            for j in range(4):
                p.rw.d[i][j] = 2 if i == j else (-1 if (i ^ j) == 2 else -3)
    ref = _run(prefix, batch, {"KMAHIP_LT_LANE": "0"}, tweak)
    _same(ref, _run(prefix, batch, {}, tweak), "transition / transversion scores")


def test_lane_classes_stand_back_when_sixteen_bits_do_not_hold_the_scores(setup):
    prefix, batch = setup

    def tweak(p):
        p.rw.W1 = -900
        p.rw.U = -300
    ref = _run(prefix, batch, {"KMAHIP_LT_LANE": "0"}, tweak)
    _same(ref, _run(prefix, batch, {}, tweak), "large gap penalties")
