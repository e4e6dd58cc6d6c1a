"""Randomised end-to-end comparison of the HIP path with the oracle: small random databases (variant families, an exact
duplicate, a tandem repeat, a poly-A island, a template shorter than a read, a reverse-complemented copy) and reads of
ragged lengths with substitutions, indels, Ns, random and chimeric reads. Stages 2, 3a, 3b, the per-read traceback and
the pile-up / consensus must all equal the oracle's."""
import numpy as np
import pytest

import oracle
from kma_amd import formats, synth

pytestmark = pytest.mark.gpu


def _db(rng, seed):
    names, seqs = synth.make_gene_db(int(rng.integers(3, 14)), int(rng.integers(1, 7)), 80, int(rng.integers(200, 900)),
                                     float(rng.choice([0.0, 0.02, 0.05, 0.08])), seed)
    names.append("dup"); seqs.append(seqs[int(rng.integers(0, len(seqs)))].copy())
    unit = rng.integers(0, 4, int(rng.integers(20, 70)), dtype=np.uint8)
    names.append("tandem"); seqs.append(np.concatenate([rng.integers(0, 4, 90, dtype=np.uint8), unit, unit, unit, rng.integers(0, 4, 70, dtype=np.uint8)]))
    names.append("polyA"); seqs.append(np.concatenate([rng.integers(0, 4, 100, dtype=np.uint8), np.zeros(50, np.uint8), rng.integers(0, 4, 100, dtype=np.uint8)]))
    names.append("short"); seqs.append(rng.integers(0, 4, int(rng.integers(40, 100)), dtype=np.uint8))
    names.append("rc_copy"); seqs.append(synth.revcomp_codes(seqs[0]).copy())
    return names, seqs


def _reads(rng, seqs, n):
    out = []
    for _ in range(n):
        kind = rng.random()
        s = seqs[int(rng.integers(0, len(seqs)))]
        L = int(rng.integers(16, 400))
        if kind < 0.06:
            r = rng.integers(0, 4, L, dtype=np.uint8)
        else:
            L = min(L, len(s))
            st = int(rng.integers(0, len(s) - L + 1))
            r = s[st:st + L].copy()
            if kind < 0.12:                      # chimera of two templates
                s2 = seqs[int(rng.integers(0, len(seqs)))]
                L2 = min(int(rng.integers(16, 150)), len(s2))
                st2 = int(rng.integers(0, len(s2) - L2 + 1))
                r = np.concatenate([r, s2[st2:st2 + L2]])
            rate = float(rng.choice([0.0, 0.005, 0.02, 0.05]))
            m = rng.random(len(r)) < rate
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            if rng.random() < 0.25:              # indels
                parts, i = [], 0
                while i < len(r):
                    j = min(len(r), i + int(rng.integers(8, 60)))
                    parts.append(r[i:j])
                    u = rng.random()
                    if u < 0.4:
                        parts.append(rng.integers(0, 4, int(rng.integers(1, 5)), dtype=np.uint8))
                    elif u < 0.8:
                        j = min(len(r), j + int(rng.integers(1, 5)))
                    i = j
                r = np.concatenate(parts)
            if rng.random() < 0.1:
                m = rng.random(len(r)) < 0.02
                r[m] = 4
        if rng.random() < 0.5:
            r = synth.revcomp_codes(r)
        out.append(np.ascontiguousarray(r[:1000]))
    return out


import os

# KMA_FUZZ_SEEDS=a:b widens the sweep (a one-off hunt; the default 12 keep the suite short)
_lo, _hi = (int(x) for x in os.environ.get("KMA_FUZZ_SEEDS", "1:13").split(":"))


@pytest.mark.parametrize("seed", list(range(_lo, _hi)))
def test_random_database_and_reads_end_to_end(tmp_path, seed):
    from kma_amd import binding
    rng = np.random.default_rng(1000 + seed)
    names, seqs = _db(rng, seed)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs, k=int(rng.choice([12, 16, 16])))
    reads = _reads(rng, seqs, 1500)
    b = formats.pack_ragged(reads)
    db = binding.KmaHipDB(prefix)
    try:
        (rc_flag, flag, T_off, T), h = db.map_se(b)
        cc = db.conclave_se(b.length, T_off, h)
        traces = db.align_trace(b, h["rc"], cc["tmpl"])
        asm = db.assemble(b, h["rc"], cc["tmpl"], traces, consensus=True)
    finally:
        db.close()
    odb = oracle.OracleDB(prefix)
    exp = odb.scan_se(b)
    for g, e, nm in zip((rc_flag, flag, T_off, T), exp, ("rc_flag", "flag", "T_off", "T")):
        assert np.array_equal(g, e), nm
    o = odb.align_se(b, *exp)
    keep = o["n_hits"] >= 0                      # the oracle does not restate strand ties (rc_copy makes some)
    assert np.array_equal(o["n_hits"][keep], h["n_hits"][keep]) and np.array_equal(o["best_score"][keep], h["best_score"][keep])
    for i in np.nonzero(keep & (o["n_hits"] > 0))[0]:
        a, c = int(T_off[i]), int(o["n_hits"][i])
        for key in ("tmpl", "start", "end", "score"):
            assert np.array_equal(o[key][a:a + c], h[key][a:a + c]), (i, key)
    # stage 3b on the device's own stage-3a result (ties included) vs the oracle on the same arrays
    tlen = formats.read_lengths(prefix)
    oc = oracle.conclave(h["n_hits"], h["best_score"], b.length, np.zeros(b.n, np.int32), T_off[:-1], h["tmpl"], h["start"], h["end"],
                         h["alignment_scores"], h["uniq_alignment_scores"], tlen)
    assert np.array_equal(cc["tmpl"], oc["tmpl"]) and np.array_equal(cc["w_scores"], oc["w_scores"])
    # stage 3c: every read against its template, then pile-up + consensus per template in the reference's order
    al = oracle.OracleAligner(odb)
    stats, off, nops, ops = traces
    per_t = {}
    for i in range(b.n - 1, -1, -1):
        tt = int(cc["tmpl"][i])
        if not tt:
            assert not stats[i].any()
            continue
        rd = reads[i]
        if (int(h["rc"][i]) & 1) != (tt < 0):
            rd = synth.revcomp_codes(rd)
        tr = al.align_trace(rd, abs(tt))
        if tr is None:
            assert not stats[i].any(), i
            continue
        cig = binding.cigar_from_runs(ops[off[i]:off[i] + nops[i]], int(stats[i][4]), int(stats[i][5]))
        assert (tr["score"], tr["start"], tr["end"], tr["aln_len"], tr["mapQ"], tr["cigar"]) == \
               (int(stats[i][0]), int(stats[i][1]), int(stats[i][2]), int(stats[i][3]), int(stats[i][9]), cig), i
        per_t.setdefault(abs(tt), []).append((tr, rd))
    for t, lst in per_t.items():
        oa = oracle.Assembly(odb, t, tlen[t])
        for tr, rd in lst:
            oa.add(tr, rd)
        call = oa.call()
        assert (call["cover"], call["aln_len"], call["depth"], call["asm_len"]) == \
               (asm["cover"][t], asm["aln_len"][t], asm["depth"][t], asm["asm_len"][t]), t
        assert call["consensus"] == asm["consensus"][t], t
