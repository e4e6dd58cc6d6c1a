"""Stage 3b (ConClave + `.res` row statistics) on the GPU vs the oracle and vs the reference's own `.res` / `.frag.gz`."""
import numpy as np
import pytest

import golden_util
import oracle
from kma_amd import formats

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hipdb_for():
    from kma_amd import binding
    opened = []

    def make(prefix):
        db = binding.KmaHipDB(prefix)
        opened.append(db)
        return db
    yield make
    for db in opened:
        db.close()


def _rows_as_stats(rows, D):
    st = dict(expected=np.zeros(D), q_value=np.zeros(D), p_value=np.ones(D), significant=np.zeros(D, np.int32))
    for r in rows:
        st["expected"][r.template_id] = r.expected
        st["q_value"][r.template_id] = r.q_value
        st["p_value"][r.template_id] = r.p_value
        st["significant"][r.template_id] = r.significant
    return st


def _se_case(g, name, hipdb_for):
    db = hipdb_for(g["prefix"])
    b = g["batch"]
    (rc_flag, flag, T_off, T), h = db.map_se(b)
    cc = db.conclave_se(b.length, T_off, h)
    tlen = formats.read_lengths(g["prefix"])
    # oracle on the same stage-3a result
    oc = oracle.conclave(h["n_hits"], h["best_score"], b.length, np.zeros(b.n, np.int32), T_off[:-1], h["tmpl"], h["start"], h["end"],
                         h["alignment_scores"], h["uniq_alignment_scores"], tlen)
    for key in ("tmpl", "start", "end", "w_scores", "depth"):
        assert np.array_equal(cc[key], oc[key]), key
    assert np.array_equal(cc["fragment_counts"], oc["fragmentCounts"]) and np.array_equal(cc["read_counts"], oc["readCounts"])
    rows = db.res_rows(cc["w_scores"])
    ost = oracle.res_stats(cc["w_scores"], tlen)
    st = _rows_as_stats(rows, len(tlen))
    assert [r.template_id for r in rows] == [t for t in range(1, len(tlen)) if cc["w_scores"][t] > 0]
    for key in ("expected", "q_value", "p_value", "significant"):
        assert np.array_equal(st[key][cc["w_scores"] > 0], ost[key][cc["w_scores"] > 0]), key
    # and against the files the reference wrote
    hdrs = [r["hdr"].rstrip(b"\0").decode() for r in g["s1"]]
    return golden_util.check_conclave_against_outputs(name, hdrs, h["n_hits"], cc["tmpl"], cc["w_scores"], st, tlen)


def test_conclave_matches_reference_res_and_frags(golden_se, hipdb_for):
    seen, rows = _se_case(golden_se, "se", hipdb_for)
    assert seen > 900 and rows > 50


def test_conclave_matches_reference_res_long_reads(golden_long, hipdb_for):
    seen, rows = _se_case(golden_long, "long", hipdb_for)
    assert rows > 0


def test_conclave_paired_matches_reference_res(golden_pe, hipdb_for):
    """Record slots of map_pe -> ConClave; a proper pair scores once, the empty-list records inherit the previous
    record's first hit (conclave.c:123-127). Reads stage 1 emitted singly go through the single-end calls; the two
    result sets share the ConClave vectors."""
    import pe_util
    g = golden_pe
    db = hipdb_for(g["prefix"])
    tlen = formats.read_lengths(g["prefix"])
    res = pe_util.hip_pe_conclave(db, g)
    st = _rows_as_stats(db.res_rows(res["w_scores"]), len(tlen))
    # oracle ConClave over the oracle's own records gives the same per-template scores
    r = pe_util.oracle_pe_conclave_records(g)
    oc = oracle.conclave(r["n_hits"], r["score"], r["q_len"], r["q_len2"], r["off"], r["tmpl"], r["start"], r["end"],
                         r["alignment_scores"], r["uniq_alignment_scores"], tlen)
    assert np.array_equal(res["w_scores"], oc["w_scores"])
    assert np.array_equal(res["depth"], oc["depth"])
    seen, rows = golden_util.check_conclave_against_outputs("pe", [], [], [], res["w_scores"], st, tlen)
    assert rows > 50
