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
