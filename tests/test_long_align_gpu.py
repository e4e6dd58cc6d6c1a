"""Stage 3a for long reads: the tasks of reads over 1 kb (single end, strand decided by stage 2, no N's) go through the wavefront-per-read
pipeline of longtrace.hip in KMA_score mode (align.hip: long_tasks) instead of a lane per task. Whole runs of the C host program with the
route on (the default) and off (KMAHIP_ALIGN_LONG=0: the lane kernel, pinned against the reference binary since round 2) must write the
same three files -- in the default mode (records with query bounds) and with -1t1, on batches that mix short reads, long reads, long
reads with N's (never routed), chimeras and reads that map nowhere, exact copies with a deletion k bases before their end (where the seeding rules of KMA_score and KMA() part), against near-identical templates (ties, template choice); and with
the route's threshold lowered so that every read of 300 bases and more takes it. The reference itself is the other side in
tests/test_reference_binary_gpu.py::test_bcnano_without_mt1_equals_reference_binary, whose reads all take the route."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from kma_amd import formats, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    tmp = tmp_path_factory.mktemp("longalign")
    rng = np.random.default_rng(52)
    genes = [rng.integers(0, 4, int(rng.integers(2500, 12000)), dtype=np.uint8) for _ in range(14)]
    genes += [g.copy() for g in genes[:5]]
    for g in genes[14:]:                                   # near-copies: ties and template choice
        x = rng.random(len(g)) < 0.015
        g[x] = (g[x] + rng.integers(1, 4, int(x.sum()), dtype=np.uint8)) & 3
    genes.append(np.concatenate([genes[0][:1500], genes[0][:1500], genes[0][:1500]]))      # a tandem repeat: duplicated k-mers
    prefix = str(tmp / "db")
    synth.write_fasta(prefix + ".fsa", [f"g{i}" for i in range(len(genes))], genes)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = []
    for i in range(900):
        g = genes[int(rng.integers(0, len(genes)))]
        u = i % 10
        if u < 3:                                          # short reads between the long ones
            L = int(rng.integers(60, 900))
        else:
            L = int(rng.integers(1100, min(9000, len(g))))
        r = synth.make_long_reads(g, 1, read_len=L, seed=3000 + i)[0]
        if u == 4:                                         # a chimera
            h = genes[int(rng.integers(0, len(genes)))]
            r = np.concatenate([r, synth.make_long_reads(h, 1, read_len=1500, seed=7000 + i)[0]])
        if u == 5:                                         # N's: such a read stays with the lane kernel
            r = r.copy(); r[rng.integers(0, len(r), int(rng.integers(1, 5)))] = 4
        if u == 6 and i % 20 == 6:                         # maps nowhere
            r = rng.integers(0, 4, len(r), dtype=np.uint8)
        if u == 7:                                         # a long foreign stretch in the middle: wide problems, or a failed join
            a = int(rng.integers(200, max(201, len(r) - 800))); w = int(rng.integers(100, 700))
            r = r.copy(); r[a:a + w] = rng.integers(0, 4, min(w, len(r) - a), dtype=np.uint8)
        if u == 8 and len(g) - L - 60 > 0:
            # an exact copy with a deletion (or a jump) k = 16 bases before its end: a MEM ends with exactly k bases left in the stretch,
            # where KMA_score still seeds and KMA() does not (align.c:541 against :306)
            a = int(rng.integers(0, len(g) - L - 60)); dlt = int(rng.choice([2, 5, 40]))
            r = np.concatenate([g[a:a + L - 16], g[a + L - 16 + dlt:a + L + dlt]])
        reads.append(r)
    fq = str(tmp / "ont.fq")
    synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    return prefix, fq, tmp, genes


def _run(prefix, fq, out, flags, env):
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", out] + flags, check=True, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, **env))
    return [open(out + ".res", "rb").read(), open(out + ".fsa", "rb").read(), gzip.open(out + ".frag.gz").read()]


@pytest.mark.parametrize("flags", [["-chain", "-bcNano"], ["-1t1"], ["-1t1", "-bcNano"]], ids=["default_mode", "1t1", "1t1_bcnano"])
def test_long_reads_through_the_pipeline_equal_the_lane_kernel(data, flags):
    prefix, fq, tmp, _ = data
    lanes = _run(prefix, fq, str(tmp / "lanes"), flags, {"KMAHIP_ALIGN_LONG": "0"})
    assert lanes[2].count(b"\n") > 500
    for env in ({}, {"KMAHIP_ALIGN_LONG": "300"}, {"KMAHIP_MAP_BATCH": "97"}):
        got = _run(prefix, fq, str(tmp / "got"), flags, env)
        for a, b, what in zip(got, lanes, (".res", ".fsa", ".frag.gz")):
            assert a == b, f"{flags} {env}: {what} differs"


def test_long_reads_equal_the_reference_binary(data):
    """... and against the reference itself, in its default mode"""
    prefix, fq, tmp, _ = data
    subprocess.run([KMA, "-i", fq, "-o", str(tmp / "ref"), "-t_db", prefix, "-bcNano", "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = _run(prefix, fq, str(tmp / "gotr"), ["-chain", "-bcNano"], {})
    assert got[0] == open(tmp / "ref.res", "rb").read()
    assert got[1] == open(tmp / "ref.fsa", "rb").read()
    assert got[2] == gzip.open(tmp / "ref.frag.gz").read()


def test_stage_3a_hits_of_long_reads_equal_the_lane_kernel(data):
    """kmahip_map_se (stages 2 + 3a) on long reads that hang over the ends of their genes by a few junk bases and begin or end a few
    bases inside them: leadTailAln / trailTailAln then open with gap columns, which KMA() trims away and KMA_score keeps (no
    Frag_align: align.c:95-118, 174-198) -- start, end and the normalised score of a hit depend on it. Every per-read and per-candidate
    figure of the route must equal the lane kernel's."""
    prefix, _, _, genes = data
    from kma_amd import binding
    rng = np.random.default_rng(99)
    reads = []
    for i in range(400):
        g = genes[int(rng.integers(0, len(genes)))]
        a = int(rng.integers(0, 12)); b = len(g) - int(rng.integers(0, 12))
        if i % 3 == 0:                                     # inside the gene on one side
            a = int(rng.integers(0, len(g) // 2))
        r = synth.make_long_reads(g[a:b], 1, read_len=b - a, seed=11000 + i)[0] if i % 2 else g[a:b].copy()
        if len(r) > b - a:
            r = r[:b - a]
        junk = lambda: rng.integers(0, 4, int(rng.integers(0, 90)), dtype=np.uint8)
        r = np.concatenate([junk(), r, junk()])
        if rng.random() < 0.5:
            r = synth.revcomp_codes(r)
        reads.append(r)
    batch = formats.pack_ragged(reads)

    def run(env):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        db = binding.KmaHipDB(prefix)
        try:
            (rc_flag, flag, T_off, T), h = db.map_se(batch)
            return T_off.copy(), T.copy(), {k: v.copy() for k, v in h.items()}
        finally:
            db.close()
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    lanes, got = run({"KMAHIP_ALIGN_LONG": "0"}), run({})
    assert np.array_equal(lanes[0], got[0]) and np.array_equal(lanes[1], got[1])
    assert int((lanes[2]["n_hits"] > 0).sum()) > 300
    for key in ("n_hits", "best_score", "flag", "rc", "alignment_scores", "uniq_alignment_scores"):
        assert np.array_equal(lanes[2][key], got[2][key]), key
    # (per-candidate columns: the first n_hits entries of a read's stretch)
    for r in range(len(reads)):
        o, m = int(lanes[0][r]), int(lanes[2]["n_hits"][r])
        for key in ("tmpl", "score", "start", "end"):
            assert np.array_equal(lanes[2][key][o:o + m], got[2][key][o:o + m]), (r, key)
