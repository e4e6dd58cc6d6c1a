"""Unpack the committed golden fixtures (tests/golden/) into a temp dir."""
import gzip
import lzma
import os
import shutil

import numpy as np

from kma_amd import formats

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _gunzip(path):
    with gzip.open(path, "rb") as f:
        return f.read()


def reads_from_s1(recs):
    reads = []
    for r in recs:
        codes = formats.unpack_words(r["seq"], r["seqlen"]).copy()
        if len(r["N"]):
            codes[r["N"]] = 4
        reads.append(codes)
    return reads


def load_se(tmp, name="se"):
    src = os.path.join(GOLD, name)
    tmp = str(tmp)
    prefix = os.path.join(tmp, "db")
    with lzma.open(os.path.join(src, "db.comp.b.xz"), "rb") as f, open(prefix + ".comp.b", "wb") as g:
        shutil.copyfileobj(f, g)
    for ext in (".length.b", ".seq.b", ".name"):
        shutil.copy(os.path.join(src, "db" + ext), prefix + ext)
    s1 = formats.parse_s1(_gunzip(os.path.join(src, "s1.bin.gz")))
    s2, n2 = formats.parse_s2(_gunzip(os.path.join(src, "s2.bin.gz")))
    s2x = None
    if os.path.exists(os.path.join(src, "s2_ex.bin.gz")):
        s2x, _ = formats.parse_s2(_gunzip(os.path.join(src, "s2_ex.bin.gz")))
    reads = reads_from_s1(s1)
    return dict(dir=src, prefix=prefix, s1=s1, s2=s2, s2_ex=s2x, n_reads=n2, reads=reads,
                batch=formats.pack_ragged(reads))


def check_scan_against_s2(s1, s2, rc_flag, flag, T_off, T):
    """Stage-2 result arrays vs the reference S2 tap, keyed by header."""
    exp = {r["hdr"]: r for r in s2}
    assert len(exp) == len(s2), "duplicate headers in fixture"
    seen = 0
    for i, r in enumerate(s1):
        got_T = T[T_off[i]:T_off[i + 1]]
        e = exp.get(r["hdr"])
        if e is None:
            assert len(got_T) == 0, f"read {r['hdr']} mapped but reference dropped it"
            continue
        seen += 1
        assert len(got_T) > 0, f"read {r['hdr']} unmapped but reference mapped it"
        assert e["rc_flag"] == rc_flag[i], (r["hdr"], e["rc_flag"], rc_flag[i])
        assert e["flag"] == flag[i], (r["hdr"], e["flag"], flag[i])
        assert np.array_equal(e["T"], got_T), (r["hdr"], e["T"], got_T)
    assert seen == len(s2)
    return seen


def load_frag_raw(name="se"):
    """frag_raw text tap (frags.c:64): header -> (nHits, score, starts, ends, templates)."""
    out = {}
    with gzip.open(os.path.join(GOLD, name, "out.frag_raw.gz"), "rt") as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            out[c[6]] = (int(c[1]), int(c[2]), [int(x) for x in c[3].split(",")],
                         [int(x) for x in c[4].split(",")], [int(x) for x in c[5].split(",")])
    return out


def check_align_against_frag_raw(s1, frag, T_off, res):
    """Stage-3a result arrays vs the reference frag_raw tap."""
    n_ok = 0
    for i, r in enumerate(s1):
        h = r["hdr"].rstrip(b"\0").decode()
        nh, o = int(res["n_hits"][i]), int(T_off[i])
        got = None
        if nh > 0:
            got = (nh, int(res["best_score"][i]), res["start"][o:o + nh].tolist(), res["end"][o:o + nh].tolist(),
                   res["tmpl"][o:o + nh].tolist())
        assert nh >= 0, f"{h}: strand tie not handled"
        assert got == frag.get(h), (h, frag.get(h), got)
        n_ok += got is not None
    assert n_ok == len(frag)
    return n_ok


def s2_record_bytes(seqlen, words, N, rc_flag, T, hdr, flag):
    """One S2 record exactly as print_ankers writes it (ankers.c:30-50)."""
    import struct
    b = struct.pack("<7i", seqlen, len(words), len(N), rc_flag, len(T), len(hdr), flag)
    return b + np.asarray(words, np.uint64).tobytes() + np.asarray(N, np.int32).tobytes() + \
        np.asarray(T, np.int32).tobytes() + hdr


def load_pe(tmp, name="pe"):
    src = os.path.join(GOLD, name)
    tmp = str(tmp)
    prefix = os.path.join(tmp, "db")
    with lzma.open(os.path.join(src, "db.comp.b.xz"), "rb") as f, open(prefix + ".comp.b", "wb") as g:
        shutil.copyfileobj(f, g)
    for ext in (".length.b", ".seq.b", ".name"):
        shutil.copy(os.path.join(src, "db" + ext), prefix + ext)
    s1 = formats.parse_s1(_gunzip(os.path.join(src, "s1.bin.gz")))
    s2_bytes = _gunzip(os.path.join(src, "s2.bin.gz"))
    # units: ("pe", i, i+1) or ("se", i)
    units, i = [], 0
    while i < len(s1):
        if s1[i]["pair"]:
            units.append(("pe", i, i + 1)); i += 2
        else:
            units.append(("se", i)); i += 1
    return dict(dir=src, prefix=prefix, s1=s1, s2_bytes=s2_bytes, units=units)


def pe_stream_from(g, scan_pair, scan_single, rc_packed):
    """Rebuild the S2 byte stream of a mixed PE/SE S1 input from per-unit results.
    scan_pair(a, b) -> list of record dicts (mate, rc, rc_flag, flag, T) in stream order;
    scan_single(r) -> None or (rc_flag, flag, T)."""
    import struct
    out = b""
    for u in g["units"]:
        if u[0] == "se":
            r = g["s1"][u[1]]
            res = scan_single(r)
            if res is not None:
                rf, fl, T = res
                words, N = r["seq"], r["N"]
                if fl & 16:
                    words, N = rc_packed(r["seq"], r["seqlen"], r["N"])
                out += s2_record_bytes(r["seqlen"], words, N, rf, T, r["hdr"], fl)
        else:
            a, b = g["s1"][u[1]], g["s1"][u[2]]
            for rec in scan_pair(a, b):
                r = (a, b)[rec["mate"]]
                words, N = r["seq"], r["N"]
                if rec["rc"]:
                    words, N = rc_packed(r["seq"], r["seqlen"], r["N"])
                out += s2_record_bytes(r["seqlen"], words, N, int(rec["rc_flag"]), rec["T"], r["hdr"], int(rec["flag"]))
    return out + struct.pack("<i", -len(g["units"]))


def load_res(name="se"):
    """`.res` rows of the full reference run: template name -> (Score, Expected, Template_length, q_value, p_value) with
    the last two kept as printed ("%8.2f", "%4.1e", runkma.c:809)."""
    out = {}
    with open(os.path.join(GOLD, name, "out.res")) as f:
        for line in f:
            if line.startswith("#"):
                continue
            c = [x.strip() for x in line.rstrip("\n").split("\t")]
            out[c[0]] = (int(c[1]), int(c[2]), int(c[3]), c[9], c[10])
    return out


def load_res_identity(name="se"):
    """template name -> (Template_Identity, Template_Coverage, Query_Identity, Query_Coverage, Depth) as printed"""
    out = {}
    with open(os.path.join(GOLD, name, "out.res")) as f:
        for line in f:
            if line.startswith("#"):
                continue
            c = [x.strip() for x in line.rstrip("\n").split("\t")]
            out[c[0]] = tuple(c[4:9])
    return out


def load_frags(name="se"):
    """`.frag.gz` rows: header -> (number of equally good templates, template name). Score / start / end of these rows come
    from the stage-3c re-alignment (assembly.c:1940-1965), not from ConClave, and are not used."""
    out = {}
    if not os.path.exists(os.path.join(GOLD, name, "out.frag.gz")):
        return None
    with gzip.open(os.path.join(GOLD, name, "out.frag.gz"), "rt") as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            out[c[6]] = (int(c[1]), c[5])
    return out


def template_names(name="se"):
    with open(os.path.join(GOLD, name, "db.name")) as f:
        return [l.rstrip("\n") for l in f]


def check_conclave_against_outputs(name, headers, n_hits, picked, w_scores, stats, tlen):
    """Stage 3b vs the reference's final files: every `.frag.gz` row names the template ConClave gave the read and the
    number of templates it tied over; every `.res` row opens with Score (= w_scores), Expected, Template_length and
    closes with q_value, p_value. Rows exist only for significant templates that also pass the consensus-identity gate of
    stage 3c, so the `.res` templates must be a subset of ours."""
    names = template_names(name)
    frags = load_frags(name)
    res = load_res(name)
    seen = 0
    for i, h in enumerate(headers if frags is not None else []):
        e = frags.get(h)
        if e is None:
            continue
        seen += 1
        assert int(n_hits[i]) == e[0], (h, e, int(n_hits[i]))
        assert names[abs(int(picked[i])) - 1] == e[1], (h, e, int(picked[i]))
    assert frags is None or seen == len(frags)
    rows = 0
    for t in range(1, len(tlen)):
        if w_scores[t] == 0:
            continue
        nm = names[t - 1]
        if nm in res:
            rows += 1
            assert stats["significant"][t], nm
            got = (int(w_scores[t]), int(stats["expected"][t]), int(tlen[t]), "%.2f" % stats["q_value"][t], "%4.1e" % stats["p_value"][t])
            assert got == res[nm], (nm, got, res[nm])
    assert rows == len(res)
    return seen, rows


def load_sam(name="se"):
    """Mapped SAM records of the reference run (tests/golden/make_golden_sam.py): qname -> list of
    (flag, rname, pos, mapq, cigar, AS) in output order (a pair has two records under one name)."""
    out = {}
    with gzip.open(os.path.join(GOLD, name, "out.sam.tsv.gz"), "rt") as f:
        for line in f:
            c = line.rstrip("\n").split("\t")
            out.setdefault(c[0], []).append((int(c[1]), c[2], int(c[3]), int(c[4]), c[5], int(c[6])))
    return out


def load_fsa(name="se"):
    """consensus FASTA of the reference run as text"""
    return _gunzip(os.path.join(GOLD, name, "out.fsa.gz")).decode()


def fsa_text(entries):
    """printConsensus (printconsensus.c:38-60, ref_fsa = 0): the consensus line without its '-' columns, 60 per line"""
    out = []
    for nm, cons in entries:
        q = cons.replace("-", "")
        out.append(">" + nm + "\n")
        out += [q[i:i + 60] + "\n" for i in range(0, len(q), 60)]
    return "".join(out)


def load_mt1(tmp):
    """tests/golden/mt1 (make_golden_mt1.py): index unpacked into tmp, the raw reads (codes 0-4, file order) with their names"""
    src = os.path.join(GOLD, "mt1")
    tmp = str(tmp)
    os.makedirs(tmp, exist_ok=True)
    prefix = os.path.join(tmp, "db")
    with lzma.open(os.path.join(src, "db.comp.b.xz"), "rb") as f, open(prefix + ".comp.b", "wb") as g:
        shutil.copyfileobj(f, g)
    for ext in (".length.b", ".seq.b", ".name"):
        shutil.copy(os.path.join(src, "db" + ext), prefix + ext)
    lut = np.full(256, 4, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    names, reads = [], []
    lines = _gunzip(os.path.join(src, "reads.fq.gz")).split(b"\n")
    for i in range(0, len(lines) - 3, 4):
        names.append(lines[i][1:].decode())
        reads.append(lut[np.frombuffer(lines[i + 1], np.uint8)])
    fq = os.path.join(tmp, "reads.fq")
    with open(fq, "wb") as f:
        f.write(b"\n".join(lines))
    return dict(dir=src, prefix=prefix, names=names, reads=reads, fastq=fq)


def load_frag_rows(name):
    """`.frag.gz` rows in file order: (sequence, n_templates, score, start, end, template name, header)"""
    with gzip.open(os.path.join(GOLD, name, "out.frag.gz"), "rt") as f:
        return [tuple(line.rstrip("\n").split("\t")) for line in f]
