"""Shared helpers of the paired-end tests: expected frag_raw lines from the oracle, tap parsing."""
import collections
import gzip
import os

import numpy as np

import golden_util
from kma_amd import formats


def load_frag_raw_lines(name):
    out = []
    with gzip.open(os.path.join(golden_util.GOLD, name, "out.frag_raw.gz"), "rt") as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            out.append((c[6], int(c[1]), int(c[2]), [int(x) for x in c[3].split(",")], [int(x) for x in c[4].split(",")],
                        [int(x) for x in c[5].split(",")]))
    return out


def compare_lines(exp, tap):
    """exp entries: (hdr, n, score, starts, ends, tmpls) or (hdr, None) = branch whose text tap is not reliable
    (the reference prints score arrays as templates there). With n == 0 the tap prints a stale first element."""
    for i, (e, l) in enumerate(zip(exp, tap)):
        assert e[0] == l[0], (i, e, l)
        if e[1] is None:
            continue
        if e[1] == 0:
            assert (e[1], e[2]) == (l[1], l[2]), (i, e, l)
        else:
            assert tuple(e) == tuple(l), (i, e, l)


def emitted(r, rc):
    import oracle
    return oracle.rc_packed(r["seq"], r["seqlen"], r["N"]) if rc else (r["seq"], r["N"])


def codes_of(words, seqlen, N):
    c = formats.unpack_words(np.asarray(words), seqlen).copy()
    if len(N):
        c[np.asarray(N)] = 4
    return c


def oracle_pe_lines(g):
    import oracle
    db = oracle.OracleDB(g["prefix"])
    al = oracle.OracleAligner(db)
    exp, kinds = [], collections.Counter()

    def se_record(r, rc, rc_flag, flag, T):
        words, N = emitted(r, rc)
        b = formats.pack_ragged([codes_of(words, r["seqlen"], N)])
        res = db.align_se(b, np.array([rc_flag], np.int32), np.array([flag & ~16], np.int32),
                          np.array([0, len(T)], np.int64), np.asarray(T, np.int32))
        nh = int(res["n_hits"][0])
        if nh > 0:
            exp.append((r["hdr"].rstrip(b"\0").decode(), nh, int(res["best_score"][0]), res["start"][:nh].tolist(),
                        res["end"][:nh].tolist(), res["tmpl"][:nh].tolist()))

    for u in g["units"]:
        if u[0] == "se":
            r = g["s1"][u[1]]
            rf, fl, To, T = db.scan_se(formats.pack_ragged([codes_of(r["seq"], r["seqlen"], r["N"])]))
            if To[1] > To[0]:
                se_record(r, int(fl[0]) & 16, int(rf[0]), int(fl[0]), T)
            continue
        a, b = g["s1"][u[1]], g["s1"][u[2]]
        _, recs = db.scan_pe(a["seq"], a["seqlen"], a["N"], b["seq"], b["seqlen"], b["N"])
        if len(recs) == 2 and len(recs[0]["T"]) == 0:
            ra, rb = (a, b)[recs[0]["mate"]], (a, b)[recs[1]["mate"]]
            wa, na = emitted(ra, recs[0]["rc"])
            wb, nb = emitted(rb, recs[1]["rc"])
            _, o = al.align_pe(np.asarray(wa), ra["seqlen"], np.asarray(na), recs[0]["flag"], np.asarray(wb), rb["seqlen"],
                               np.asarray(nb), recs[1]["flag"], recs[1]["T"])
            kinds[o["kind"]] += 1
            ha, hb = ra["hdr"].rstrip(b"\0").decode(), rb["hdr"].rstrip(b"\0").decode()
            if o["kind"] == 1:
                n = o["n_hits"]
                row = (n, o["best"], o["start"][:n].tolist(), o["end"][:n].tolist(), o["tmpl"][:n].tolist())
                exp.append((ha,) + row)
                exp.append((hb,) + row)
            elif o["kind"] == 2:
                exp += [(ha, None), (hb, None)]
            elif o["kind"] == 3:
                exp.append((ha, None))
            elif o["kind"] == 4:
                exp.append((hb, None))
        else:
            for rec in recs:
                assert bool(rec["rc"]) == bool(rec["flag"] & 16)
                se_record((a, b)[rec["mate"]], rec["rc"], rec["rc_flag"], rec["flag"], rec["T"])
    return exp, kinds


def oracle_pe_conclave_records(g):
    """The frag_raw records of the paired run as ConClave input arrays: one record per proper pair (score negated,
    updatescores.c:470-488), one per singly written read; plus the two ConClave score vectors."""
    import oracle
    db = oracle.OracleDB(g["prefix"])
    al = oracle.OracleAligner(db)
    rec = []        # (n_hits, signed score, q_len, q_len2, tmpl, start, end)
    vec = [np.zeros(len(al.alignment_scores), np.uint64), np.zeros(len(al.alignment_scores), np.uint64)]

    def se_record(r, rc, rc_flag, flag, T):
        words, N = emitted(r, rc)
        b = formats.pack_ragged([codes_of(words, r["seqlen"], N)])
        res = db.align_se(b, np.array([rc_flag], np.int32), np.array([flag & ~16], np.int32),
                          np.array([0, len(T)], np.int64), np.asarray(T, np.int32))
        vec[0] += res["alignment_scores"]; vec[1] += res["uniq_alignment_scores"]
        nh = int(res["n_hits"][0])
        if nh > 0:
            rec.append((nh, int(res["best_score"][0]), r["seqlen"], 0, res["tmpl"][:nh].tolist(), res["start"][:nh].tolist(),
                        res["end"][:nh].tolist()))

    for u in g["units"]:
        if u[0] == "se":
            r = g["s1"][u[1]]
            rf, fl, To, T = db.scan_se(formats.pack_ragged([codes_of(r["seq"], r["seqlen"], r["N"])]))
            if To[1] > To[0]:
                se_record(r, int(fl[0]) & 16, int(rf[0]), int(fl[0]), T)
            continue
        a, b = g["s1"][u[1]], g["s1"][u[2]]
        _, recs = db.scan_pe(a["seq"], a["seqlen"], a["N"], b["seq"], b["seqlen"], b["N"])
        if len(recs) == 2 and len(recs[0]["T"]) == 0:
            ra, rb = (a, b)[recs[0]["mate"]], (a, b)[recs[1]["mate"]]
            wa, na = emitted(ra, recs[0]["rc"])
            wb, nb = emitted(rb, recs[1]["rc"])
            _, o = al.align_pe(np.asarray(wa), ra["seqlen"], np.asarray(na), recs[0]["flag"], np.asarray(wb), rb["seqlen"],
                               np.asarray(nb), recs[1]["flag"], recs[1]["T"])
            n, nr = o["n_hits"], o["n_hits_r"]
            cut = lambda lo, hi: (o["tmpl"][lo:hi].tolist(), o["start"][lo:hi].tolist(), o["end"][lo:hi].tolist())
            if o["kind"] == 1:
                rec.append((n, -o["best"], ra["seqlen"], rb["seqlen"]) + cut(0, n))      # n == 0: written with an empty list
            elif o["kind"] == 2:
                if n:
                    rec.append((n, o["best"], ra["seqlen"], 0) + cut(0, n))
                if nr:
                    rec.append((nr, o["best_r"], rb["seqlen"], 0) + cut(n, n + nr))
            elif o["kind"] == 3 and n:
                rec.append((n, o["best"], ra["seqlen"], 0) + cut(0, n))
            elif o["kind"] == 4 and n:
                rec.append((n, o["best_r"], rb["seqlen"], 0) + cut(0, n))
        else:
            for x in recs:
                se_record((a, b)[x["mate"]], x["rc"], x["rc_flag"], x["flag"], x["T"])
    vec[0] += al.alignment_scores; vec[1] += al.uniq_alignment_scores
    off = np.concatenate([[0], np.cumsum([r[0] for r in rec])]).astype(np.int64)
    flat = lambda i: np.array([x for r in rec for x in r[i]], np.int32)
    col = lambda i: np.array([r[i] for r in rec], np.int32)
    return dict(n_hits=col(0), score=col(1), q_len=col(2), q_len2=col(3), off=off[:-1], tmpl=flat(4), start=flat(5), end=flat(6),
                alignment_scores=vec[0], uniq_alignment_scores=vec[1])


def hip_pe_conclave(db, g):
    """Stages 2 + 3a on the device for the pairs (map_pe) and the singly emitted reads (map_se) of a paired fixture, the two
    results merged into frag_raw records in stream order, then stage 3b over them (KmaHipDB.conclave_records)."""
    codes = lambda r: codes_of(r["seq"], r["seqlen"], r["N"])
    pairs = [u for u in g["units"] if u[0] == "pe"]
    singles = [u for u in g["units"] if u[0] == "se"]
    pb = formats.pack_ragged([codes(g["s1"][i]) for u in pairs for i in (u[1], u[2])])
    (mate, rc, rc_flag, flag, R_off, T), h = db.map_pe(pb)
    sb = formats.pack_ragged([codes(g["s1"][u[1]]) for u in singles])
    (_, _, sT_off, _), sh = db.map_se(sb)
    pj = {u[1]: j for j, u in enumerate(pairs)}
    sj = {u[1]: j for j, u in enumerate(singles)}
    rec, frags = [], []      # frags[k] = fragments of record k in record order: (index into g["s1"], stage-3a flag, rc)

    def add(n, score, ql, ql2, src, o, fr):
        rec.append((n, score, ql, ql2, src["tmpl"][o:o + n].tolist(), src["start"][o:o + n].tolist(), src["end"][o:o + n].tolist()))
        frags.append(fr)

    for u in g["units"]:
        if u[0] == "se":
            j = sj[u[1]]
            if sh["n_hits"][j] > 0:
                add(int(sh["n_hits"][j]), int(sh["best_score"][j]), int(sb.length[j]), 0, sh, int(sT_off[j]), [(u[1], int(sh["flag"][j]), int(sh["rc"][j]))])
            continue
        j = pj[u[1]]
        r0, r1 = 2 * j, 2 * j + 1
        kind = int(h["kind"][j])
        ln = lambda x: int(pb.length[2 * j + int(mate[x])])
        fg = lambda x: ((u[1], u[2])[int(mate[x])], int(h["flag"][x]), int(h["rc"][x]) & 1)
        o = int(R_off[r1])
        if kind == 1:
            swapped = int(h["rc"][r1]) & 2        # the second slot's fragment is written first (alnfrags.c:1807-1812)
            add(int(h["n_hits"][r1]), -int(h["best_score"][r1]), ln(r1) if swapped else ln(r0), ln(r0) if swapped else ln(r1), h, o,
                [fg(r1), fg(r0)] if swapped else [fg(r0), fg(r1)])
        elif kind == 2:
            n0, n1 = int(h["n_hits"][r0]), int(h["n_hits"][r1])
            add(n0, int(h["best_score"][r0]), ln(r0), 0, h, o, [fg(r0)])
            add(n1, int(h["best_score"][r1]), ln(r1), 0, h, o + n0, [fg(r1)])
        elif kind in (3, 4):
            x = r0 if kind == 3 else r1
            add(int(h["n_hits"][x]), int(h["best_score"][x]), ln(x), 0, h, o, [fg(x)])
        else:
            for x in (r0, r1):
                if mate[x] >= 0 and h["n_hits"][x] > 0:
                    add(int(h["n_hits"][x]), int(h["best_score"][x]), ln(x), 0, h, int(R_off[x]), [fg(x)])
    off = np.concatenate([[0], np.cumsum([r[0] for r in rec])]).astype(np.int64)
    flat = lambda i: np.array([x for r in rec for x in r[i]], np.int32)
    col = lambda i: np.array([r[i] for r in rec], np.int32)
    AS = h["alignment_scores"] + sh["alignment_scores"]
    US = h["uniq_alignment_scores"] + sh["uniq_alignment_scores"]
    out = db.conclave_records(col(0), col(1), col(2), col(3), off, flat(4), flat(5), flat(6), AS, US)
    # the slot form on the pairs alone must agree with the records form over the same pairs
    vec = dict(h); vec["alignment_scores"], vec["uniq_alignment_scores"] = AS, US
    slot = db.conclave_pe(pb.length, mate, R_off, vec)
    prec = [(n, sc, a, b2, t, s, e) for (n, sc, a, b2, t, s, e), is_pair in zip(rec, _pair_record_mask(g, pj, sj, h, sh, mate)) if is_pair]
    poff = np.concatenate([[0], np.cumsum([r[0] for r in prec])]).astype(np.int64)
    pflat = lambda i: np.array([x for r in prec for x in r[i]], np.int32)
    pcol = lambda i: np.array([r[i] for r in prec], np.int32)
    only = db.conclave_records(pcol(0), pcol(1), pcol(2), pcol(3), poff, pflat(4), pflat(5), pflat(6), AS, US)
    assert np.array_equal(slot["w_scores"], only["w_scores"]) and np.array_equal(slot["depth"], only["depth"])
    assert np.array_equal(slot["read_counts"], only["read_counts"])
    out["frags"] = frags
    out["n_hits"] = col(0)
    return out


def _pair_record_mask(g, pj, sj, h, sh, mate):
    """For every record hip_pe_conclave builds (same order): does it come from a pair unit?"""
    mask = []
    for u in g["units"]:
        if u[0] == "se":
            if sh["n_hits"][sj[u[1]]] > 0:
                mask.append(False)
            continue
        j = pj[u[1]]
        kind = int(h["kind"][j])
        if kind == 1:
            mask.append(True)
        elif kind == 2:
            mask += [True, True]
        elif kind in (3, 4):
            mask.append(True)
        else:
            mask += [True for x in (2 * j, 2 * j + 1) if mate[x] >= 0 and h["n_hits"][x] > 0]
    return mask
