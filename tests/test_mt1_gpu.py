"""BASELINE config C4 (`-Mt1 t -bcNano`, long reads against one template) on the GPU: the long-read trace pipeline
(longtrace.hip) through the C-ABI against the reference's own SAM / .res / consensus / .frag.gz (tests/golden/mt1) and the oracle."""
import numpy as np
import pytest

import golden_util
from kma_amd import formats, synth

pytestmark = pytest.mark.gpu


def _trace(g):
    from kma_amd import binding
    db = binding.KmaHipDB(g["prefix"])
    try:
        batch = formats.pack_ragged(g["reads"])
        return db, batch, db.align_trace_mt1(batch, 1)
    except Exception:
        db.close()
        raise


def test_mt1_traceback_matches_reference_sam(tmp_path):
    from kma_amd import binding
    g = golden_util.load_mt1(tmp_path / "mt1")
    db, batch, ((stats, off, nops, ops), rc) = _trace(g)
    db.close()
    sam = golden_util.load_sam("mt1")
    mapped = 0
    for i, nm in enumerate(g["names"]):
        flag, rname, pos, mapq, cigar, AS = sam[nm][0]
        if stats[i, 3] == 0:
            assert cigar == "*", nm
            continue
        mapped += 1
        got = (int(stats[i, 1]) + 1, min(254, int(stats[i, 9])), binding.cigar_from_runs(ops[off[i]:off[i] + nops[i]], int(stats[i, 4]), int(stats[i, 5])), int(stats[i, 0]))
        assert got == (pos, mapq, cigar, AS), (nm, got[:2], got[3], (pos, mapq, AS), got[2][:80], cigar[:80])
    assert mapped == 174
    assert int(rc.sum()) > 50


def test_mt1_run_matches_reference_res_consensus_and_frags(tmp_path):
    """kmahip_run_mt1 (strand, traceback, stream-order pile-up, nanoCaller with significantAnd90Nuc) + the `.frag.gz` writer:
    the reference's `.res` row, consensus FASTA and fragment rows byte for byte."""
    import gzip
    import os
    from kma_amd import binding
    g = golden_util.load_mt1(tmp_path / "mt1")
    batch = formats.pack_ragged(g["reads"])
    db = binding.KmaHipDB(g["prefix"])
    try:
        o = db.run_mt1(batch, 1)
        names = [nm.encode() for nm in g["names"]]
        rows = db.frag_write2(str(tmp_path / "out.frag.gz"), batch, o["rc"], o["tmpl"], o["n_hits"], o["trace_stats"], names, order=1)
    finally:
        db.close()
    assert rows == 174
    got = gzip.open(tmp_path / "out.frag.gz", "rt").read()
    exp = gzip.open(os.path.join(golden_util.GOLD, "mt1", "out.frag.gz"), "rt").read()
    assert got == exp
    assert golden_util.fsa_text([("genome60k", o["consensus"][1])]) == golden_util.load_fsa("mt1")
    line = binding.KmaHipDB.res_line("genome60k", o["row"], o["cover"][1], o["aln_len"][1], o["depth"][1])
    exp_res = open(os.path.join(golden_util.GOLD, "mt1", "out.res")).read().splitlines()
    assert o["row"].significant == 1 and line.rstrip("\n") == exp_res[1]
