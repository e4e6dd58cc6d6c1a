"""Paired-end stage 2 (-apm p) on the GPU: rebuilt S2 byte stream vs the reference tap, and vs the oracle."""
import numpy as np
import pytest

import golden_util
from kma_amd import formats

pytestmark = pytest.mark.gpu


def _codes(r):
    c = formats.unpack_words(r["seq"], r["seqlen"]).copy()
    if len(r["N"]):
        c[r["N"]] = 4
    return c


def _device_results(g, db):
    pairs = [u for u in g["units"] if u[0] == "pe"]
    singles = [u for u in g["units"] if u[0] == "se"]
    pb = formats.pack_ragged([_codes(g["s1"][i]) for u in pairs for i in (u[1], u[2])])
    mate, rc, rc_flag, flag, R_off, T = db.scan_pe(pb)
    pair_res = {}
    for j, u in enumerate(pairs):
        recs = []
        for x in (2 * j, 2 * j + 1):
            if mate[x] >= 0:
                recs.append(dict(mate=int(mate[x]), rc=int(rc[x]), rc_flag=int(rc_flag[x]), flag=int(flag[x]),
                                 T=T[R_off[x]:R_off[x + 1]]))
        pair_res[u[1]] = recs
    single_res = {}
    if singles:
        sb = formats.pack_ragged([_codes(g["s1"][u[1]]) for u in singles])
        rf, fl, To, Ts = db.scan_se(sb)
        for j, u in enumerate(singles):
            single_res[u[1]] = (int(rf[j]), int(fl[j]), Ts[To[j]:To[j + 1]]) if To[j + 1] > To[j] else None
    return pair_res, single_res


def test_pe_scan_matches_reference_s2_stream(golden_pe):
    import oracle
    from kma_amd import binding
    db = binding.KmaHipDB(golden_pe["prefix"])
    try:
        pair_res, single_res = _device_results(golden_pe, db)
    finally:
        db.close()
    s1 = golden_pe["s1"]
    idx = {id(r): i for i, r in enumerate(s1)}
    got = golden_util.pe_stream_from(golden_pe, lambda a, b: pair_res[idx[id(a)]], lambda r: single_res[idx[id(r)]],
                                     oracle.rc_packed)
    assert got == golden_pe["s2_bytes"]


def test_pe_scan_forced_pairing_matches_reference_s2_stream(golden_pe):
    """stage 2 of `-apm f -1t1` (save_kmers_forcePair, savekmers.c:3779-3864: pair_force_kernel, kmahip_params.apm = 2): a couple on the
    templates both mates hit on opposite strands at the best summed score, both records carrying the sum, or no record at all; the
    records stage 1 filed singly go through save_kmers as ever. Rebuilt S2 stream against tests/golden/pe/s2_force.bin.gz. (The whole
    run with forced pairing stays refused: alnFragsForcePE is not built.)"""
    import gzip
    import os
    import oracle
    from kma_amd import binding
    g = golden_pe
    db = binding.KmaHipDB(g["prefix"])
    try:
        db.params.apm = 2
        pair_res, single_res = _device_results(g, db)
    finally:
        db.close()
    idx = {id(r): i for i, r in enumerate(g["s1"])}
    got = golden_util.pe_stream_from(g, lambda a, b: pair_res[idx[id(a)]], lambda r: single_res[idx[id(r)]], oracle.rc_packed)
    want = gzip.open(os.path.join(g["dir"], "s2_force.bin.gz")).read()
    assert sum(1 for v in pair_res.values() if len(v) == 2) > 500 and got == want


@pytest.mark.parametrize("apm,tap", [(1, "s2_default.bin.gz"), (0, "s2_default_p.bin.gz")])
def test_pe_scan_in_the_default_mode_matches_reference_s2_stream(golden_pe, apm, tap):
    """the S2 stream of `kma -ipe r1 r2 [-apm p]` WITHOUT -1t1 (tests/golden/make_golden_pe_default.py): the couples through
    kmahip_scan_pe (union pairing without -apm, kma.c:206), the records that lost their mate through kmahip_scan_chain -- zero or more
    records each with flag 0 and the query bounds behind the header (savekmers.c:196-200) -- rebuilt byte for byte"""
    import gzip
    import os
    import struct
    import oracle
    from kma_amd import binding
    g = golden_pe
    want = gzip.open(os.path.join(g["dir"], tap)).read()
    db = binding.KmaHipDB(g["prefix"])
    try:
        db.params.apm = apm
        pair_res, _ = _device_results(g, db)
        singles = [u for u in g["units"] if u[0] == "se"]
        ch = db.scan_chain(formats.pack_ragged([_codes(g["s1"][u[1]]) for u in singles]))
    finally:
        db.close()
    by_single = {}
    for x in range(len(ch["read"])):
        by_single.setdefault(int(ch["read"][x]), []).append(x)
    out, j = b"", 0
    for u in g["units"]:
        if u[0] == "se":
            r = g["s1"][u[1]]
            for x in by_single.get(j, []):
                words, N = (oracle.rc_packed(r["seq"], r["seqlen"], r["N"]) if ch["emit_rc"][x] else (r["seq"], r["N"]))
                out += golden_util.s2_record_bytes(r["seqlen"], words, N, int(ch["rc_flag"][x]), ch["T"][ch["T_off"][x]:ch["T_off"][x + 1]],
                                                   r["hdr"] + b"\x00" + struct.pack("<2i", int(ch["q_start"][x]), int(ch["q_end"][x])), 0)
            j += 1
        else:
            a, b = g["s1"][u[1]], g["s1"][u[2]]
            for rec in pair_res[u[1]]:
                r = (a, b)[rec["mate"]]
                words, N = (oracle.rc_packed(r["seq"], r["seqlen"], r["N"]) if rec["rc"] else (r["seq"], r["N"]))
                out += golden_util.s2_record_bytes(r["seqlen"], words, N, int(rec["rc_flag"]), rec["T"], r["hdr"], int(rec["flag"]))
    out += struct.pack("<i", -len(g["units"]))
    assert len(ch["read"]) >= 15 and out == want


@pytest.mark.parametrize("mode", ["p", "u", "f"])
def test_pe_scan_vs_oracle_redundant_db(tmp_path, mode):
    """Wide candidate lists (overflow path) and short mates; the pairing penalty and the union pairing (pair_union_kernel against
    oracle/scan.c's orc_scan_pe_union, which the reference's default `-ipe` tap pins) and stage 2 of forced pairing (pair_force_kernel
    against orc_scan_pe_force, pinned by the `-apm f` tap)."""
    import oracle
    from kma_amd import binding, synth
    names, seqs = synth.make_gene_db(n_families=5, variants=30, len_lo=500, len_hi=900, max_div=0.03, seed=77)
    prefix = str(tmp_path / "red")
    formats.write_index(prefix, names, seqs)
    m1, m2, _ = synth.make_pairs(seqs, 1500, read_len=100, sub_rate=0.01, seed=3)
    rng = np.random.default_rng(5)
    reads = []
    for a, b in zip(m1, m2):
        if rng.random() < 0.1:
            a = a[: int(rng.integers(12, 40))]
        if rng.random() < 0.1:
            b = rng.integers(0, 4, len(b), dtype=np.uint8)
        reads += [a, b]
    batch = formats.pack_ragged(reads)
    db = binding.KmaHipDB(prefix)
    try:
        db.params.apm = {"p": 0, "u": 1, "f": 2}[mode]
        mate, rc, rc_flag, flag, R_off, T = db.scan_pe(batch)
    finally:
        db.close()
    odb = oracle.OracleDB(prefix)
    for j in range(len(reads) // 2):
        w = lambda i: batch.seq[batch.seq_off[i]:batch.seq_off[i + 1] - 1]
        Nn = lambda i: batch.N[batch.N_off[i]:batch.N_off[i + 1]]
        _, recs = odb.scan_pe(w(2 * j), int(batch.length[2 * j]), Nn(2 * j), w(2 * j + 1), int(batch.length[2 * j + 1]), Nn(2 * j + 1), union=mode == "u", force=mode == "f")
        got = []
        for x in (2 * j, 2 * j + 1):
            if mate[x] >= 0:
                got.append((int(mate[x]), int(rc[x]), int(rc_flag[x]), int(flag[x]), T[R_off[x]:R_off[x + 1]].tolist()))
        exp = [(r["mate"], r["rc"], r["rc_flag"], r["flag"], r["T"].tolist()) for r in recs]
        assert got == exp, (j, got, exp)


def test_pe_align_matches_reference_frag_raw_tap_and_oracle(golden_pe):
    """Stage 3a on the records of the pairs: proper couples (alnFragsPenaltyPE) and single records (alnFragsSE)."""
    import oracle
    import pe_util
    from kma_amd import binding
    g = golden_pe
    pairs = [u for u in g["units"] if u[0] == "pe"]
    pb = formats.pack_ragged([_codes(g["s1"][i]) for u in pairs for i in (u[1], u[2])])
    db = binding.KmaHipDB(g["prefix"])
    try:
        (mate, rc, rc_flag, flag, R_off, T), h = db.map_pe(pb)
        singles = [u for u in g["units"] if u[0] == "se"]
        sb = formats.pack_ragged([_codes(g["s1"][u[1]]) for u in singles])
        (_, _, sT_off, _), sh = db.map_se(sb)
    finally:
        db.close()
    # expected lines in stream order, from the device results
    got = []
    pj = {u[1]: j for j, u in enumerate(pairs)}
    sj = {u[1]: j for j, u in enumerate(singles)}
    for u in g["units"]:
        if u[0] == "se":
            j = sj[u[1]]
            nh, o = int(sh["n_hits"][j]), int(sT_off[j])
            if nh > 0:
                got.append((g["s1"][u[1]]["hdr"].rstrip(b"\0").decode(), nh, int(sh["best_score"][j]),
                            sh["start"][o:o + nh].tolist(), sh["end"][o:o + nh].tolist(), sh["tmpl"][o:o + nh].tolist()))
            continue
        j = pj[u[1]]
        a, b = g["s1"][u[1]], g["s1"][u[2]]
        kind = int(h["kind"][j])
        r0, r1 = 2 * j, 2 * j + 1
        hdr = lambda x: (a, b)[int(mate[x])]["hdr"].rstrip(b"\0").decode()
        o = int(R_off[r1])
        if kind == 1:
            n = int(h["n_hits"][r1])
            row = (n, int(h["best_score"][r1]), h["start"][o:o + n].tolist(), h["end"][o:o + n].tolist(), h["tmpl"][o:o + n].tolist())
            got.append((hdr(r0),) + row)
            got.append((hdr(r1),) + row)
        elif kind == 2:
            got += [(hdr(r0), None), (hdr(r1), None)]
        elif kind == 3:
            got.append((hdr(r0), None))
        elif kind == 4:
            got.append((hdr(r1), None))
        else:
            for x in (r0, r1):
                nh = int(h["n_hits"][x])
                if mate[x] >= 0 and nh > 0:
                    ox = int(R_off[x])
                    got.append((hdr(x), nh, int(h["best_score"][x]), h["start"][ox:ox + nh].tolist(),
                                h["end"][ox:ox + nh].tolist(), h["tmpl"][ox:ox + nh].tolist()))
    tap = pe_util.load_frag_raw_lines("pe")
    assert len(got) == len(tap)
    pe_util.compare_lines(got, tap)
    # and the ConClave vectors against the oracle's over the same units
    exp_lines, kinds = pe_util.oracle_pe_lines(g)
    assert len(exp_lines) == len(got)
    assert int((h["kind"] == 1).sum()) == kinds[1]
