"""Pin the CPU oracle against the reference's own stream taps (not gpu)."""
import os

import numpy as np
import pytest

import golden_util
import oracle
from kma_amd import formats


def test_s1_roundtrip_pack(golden_se):
    # our packer reproduces the reference's 2-bit words + N lists (compdna.c:99-127)
    b = golden_se["batch"]
    for i, r in enumerate(golden_se["s1"]):
        w = (r["seqlen"] + 31) // 32
        assert np.array_equal(b.seq[b.seq_off[i]:b.seq_off[i] + w], r["seq"])
        assert np.array_equal(b.N[b.N_off[i]:b.N_off[i + 1]], r["N"])


def test_oracle_scan_matches_s2_tap(golden_se):
    db = oracle.OracleDB(golden_se["prefix"])
    rc_flag, flag, T_off, T = db.scan_se(golden_se["batch"])
    n = golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2"], rc_flag, flag, T_off, T)
    assert n > 800


def test_oracle_scan_matches_s2_tap_exhaustive(golden_se):
    db = oracle.OracleDB(golden_se["prefix"])
    rc_flag, flag, T_off, T = db.scan_se(golden_se["batch"], exhaustive=1)
    golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2_ex"], rc_flag, flag, T_off, T)


def test_index_writer_semantics(golden_se, tmp_path):
    # our .comp.b writer yields the same k-mer -> template-set map and the same
    # equal-set partition as the reference's `kma index`
    import gzip
    names, seqs, cur = [], [], None
    lut = np.full(256, 255, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    with gzip.open(golden_se["dir"] + "/db.fsa.gz", "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                names.append(line[1:].decode())
            else:
                seqs.append(lut[np.frombuffer(line, np.uint8)])
    formats.write_index(str(tmp_path / "mine"), names, seqs, k=16)
    ref = formats.read_comp_b(golden_se["prefix"] + ".comp.b")
    mine = formats.read_comp_b(str(tmp_path / "mine.comp.b"))
    assert (ref.DB_size, ref.n, ref.kmersize, ref.v_index) == (mine.DB_size, mine.n, mine.kmersize, mine.v_index)
    m_ref, m_mine = formats.comp_db_mapping(ref), formats.comp_db_mapping(mine)
    assert m_ref == m_mine
    # partition equality: value_index equal <=> set equal, in both
    assert len(set(ref.value_index.tolist())) == len(set(mine.value_index.tolist())) == len(set(m_ref.values()))
    # and the oracle gives identical stage-2 output on our index
    db = oracle.OracleDB(str(tmp_path / "mine"))
    rc_flag, flag, T_off, T = db.scan_se(golden_se["batch"])
    golden_util.check_scan_against_s2(golden_se["s1"], golden_se["s2"], rc_flag, flag, T_off, T)
    for ext in (".length.b", ".seq.b"):
        assert open(golden_se["prefix"] + ext, "rb").read() == open(str(tmp_path / "mine") + ext, "rb").read()


def _oracle_align(g):
    db = oracle.OracleDB(g["prefix"])
    rc_flag, flag, T_off, T = db.scan_se(g["batch"])
    golden_util.check_scan_against_s2(g["s1"], g["s2"], rc_flag, flag, T_off, T)
    return T_off, db.align_se(g["batch"], rc_flag, flag, T_off, T)


def test_oracle_align_matches_frag_raw_tap(golden_se):
    # stage 3a: KMA_score + chainSeeds + NW_score + alnFragsSE + update_Scores vs `-a` tap
    T_off, res = _oracle_align(golden_se)
    n = golden_util.check_align_against_frag_raw(golden_se["s1"], golden_util.load_frag_raw("se"), T_off, res)
    assert n > 900


def test_oracle_align_matches_frag_raw_tap_long_reads(golden_long):
    # long reads with divergent blocks / junk flanks: exercises NW_band_score (nw.c:892-1188)
    import ctypes
    c = (ctypes.c_int64 * 4).in_dll(oracle.lib(), "orc_counters")
    before = c[1]
    T_off, res = _oracle_align(golden_long)
    n = golden_util.check_align_against_frag_raw(golden_long["s1"], golden_util.load_frag_raw("long"), T_off, res)
    assert n > 200
    assert c[1] - before > 100, "banded NW not exercised"


def test_reference_binary_accepts_our_index(golden_se, tmp_path):
    # the reference `kma` run on an index written by formats.write_index yields the
    # same S2 stream and .res as on its own `kma index` output
    import gzip
    import os
    import subprocess
    if not os.path.exists(oracle.REF_KMA):
        import pytest
        pytest.skip("oracle/_ref/kma not built")
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    names, seqs = [], []
    with gzip.open(golden_se["dir"] + "/db.fsa.gz", "rb") as f:
        for line in f:
            line = line.strip()
            if line.startswith(b">"):
                names.append(line[1:].decode())
            else:
                seqs.append(lut[np.frombuffer(line, np.uint8)])
    prefix = str(tmp_path / "mine")
    formats.write_index(prefix, names, seqs)
    fq = str(tmp_path / "r.fq")
    open(fq, "wb").write(gzip.open(golden_se["dir"] + "/reads.fq.gz", "rb").read())
    base = [oracle.REF_KMA, "-i", fq, "-o", str(tmp_path / "o"), "-t_db", prefix, "-1t1", "-t", "1"]
    s2 = subprocess.run(base + ["-s2"], capture_output=True, check=True).stdout
    assert s2 == gzip.open(golden_se["dir"] + "/s2.bin.gz", "rb").read()
    subprocess.run(base, capture_output=True, check=True)
    assert open(str(tmp_path / "o.res")).read() == open(golden_se["dir"] + "/out.res").read()


def test_oracle_pe_scan_matches_s2_tap_bytes(golden_pe):
    # paired end `-apm p`: get_kmers_for_pair + getFirstPen/getSecondBestPen/getF_Best +
    # save_kmers_penaltyPair; the rebuilt S2 stream must equal the reference's byte for byte
    db = oracle.OracleDB(golden_pe["prefix"])

    def single(r):
        codes = formats.unpack_words(r["seq"], r["seqlen"]).copy()
        codes[r["N"]] = 4
        rf, fl, To, T = db.scan_se(formats.pack_ragged([codes]))
        return (int(rf[0]), int(fl[0]), T) if To[1] > To[0] else None

    def pair(a, b):
        return db.scan_pe(a["seq"], a["seqlen"], a["N"], b["seq"], b["seqlen"], b["N"])[1]

    got = golden_util.pe_stream_from(golden_pe, pair, single, oracle.rc_packed)
    assert got == golden_pe["s2_bytes"]


def test_oracle_pe_scan_forced_pairing_matches_s2_tap_bytes(golden_pe):
    """stage 2 of `-ipe r1 r2 -apm f -1t1` (save_kmers_forcePair: oracle/scan.c orc_scan_pe_force) against the reference's `-s2` tap
    (tests/golden/pe/s2_force.bin.gz): couples or nothing; the records stage 1 filed singly through save_kmers"""
    import gzip
    db = oracle.OracleDB(golden_pe["prefix"])

    def single(r):
        codes = formats.unpack_words(r["seq"], r["seqlen"]).copy()
        codes[r["N"]] = 4
        rf, fl, To, T = db.scan_se(formats.pack_ragged([codes]))
        return (int(rf[0]), int(fl[0]), T) if To[1] > To[0] else None

    def pair(a, b):
        return db.scan_pe(a["seq"], a["seqlen"], a["N"], b["seq"], b["seqlen"], b["N"], force=True)[1]

    got = golden_util.pe_stream_from(golden_pe, pair, single, oracle.rc_packed)
    assert got == gzip.open(os.path.join(golden_pe["dir"], "s2_force.bin.gz")).read()


@pytest.mark.parametrize("union,tap", [(False, "s2_default_p.bin.gz"), (True, "s2_default.bin.gz")])
def test_oracle_pe_scan_in_the_default_mode_matches_s2_tap_bytes(golden_pe, union, tap):
    """`-ipe r1 r2 [-apm p]` WITHOUT -1t1 (tests/golden/pe/s2_default*.bin.gz, make_golden_pe_default.py): the couples by the pairing
    penalty as above or, without -apm, by the union pairing (save_kmers_unionPair, oracle/scan.c orc_scan_pe_union); a record that lost
    its mate through the chain finder (save_kmers_batch hands it to kmerScan = save_kmers_chain, savekmers.c:196-200) -- zero or more
    records each, flag 0, the query bounds behind the header; the rebuilt stream equals the reference's byte for byte"""
    import gzip
    import struct
    db = oracle.OracleDB(golden_pe["prefix"])
    want = gzip.open(os.path.join(golden_pe["dir"], tap)).read()
    out, n_chain = b"", 0
    for u in golden_pe["units"]:
        if u[0] == "se":
            r = golden_pe["s1"][u[1]]
            codes = formats.unpack_words(r["seq"], r["seqlen"]).copy()
            codes[r["N"]] = 4
            for rc_flag, emit_rc, q_start, q_end, T in db.scan_chain(formats.pack_ragged([codes]))[0]:
                words, N = (oracle.rc_packed(r["seq"], r["seqlen"], r["N"]) if emit_rc else (r["seq"], r["N"]))
                out += golden_util.s2_record_bytes(r["seqlen"], words, N, rc_flag, T, r["hdr"] + b"\x00" + struct.pack("<2i", q_start, q_end), 0)
                n_chain += 1
        else:
            a, b = golden_pe["s1"][u[1]], golden_pe["s1"][u[2]]
            for rec in db.scan_pe(a["seq"], a["seqlen"], a["N"], b["seq"], b["seqlen"], b["N"], union=union)[1]:
                r = (a, b)[rec["mate"]]
                words, N = (oracle.rc_packed(r["seq"], r["seqlen"], r["N"]) if rec["rc"] else (r["seq"], r["N"]))
                out += golden_util.s2_record_bytes(r["seqlen"], words, N, int(rec["rc_flag"]), rec["T"], r["hdr"], int(rec["flag"]))
    out += struct.pack("<i", -len(golden_pe["units"]))
    assert n_chain >= 15
    assert out == want


def test_oracle_pe_align_matches_frag_raw_tap(golden_pe):
    # stage 3a for `-ipe ... -apm p`: alnFragsPenaltyPE + update_Scores_pe (proper pairs) and alnFragsSE for
    # the records stage 2 wrote singly; compared line by line with the reference's `-a` tap.
    import pe_util
    exp, kinds = pe_util.oracle_pe_lines(golden_pe)
    tap = pe_util.load_frag_raw_lines("pe")
    assert len(exp) == len(tap)
    assert kinds[1] > 500
    pe_util.compare_lines(exp, tap)


def _oracle_conclave_se(g):
    db = oracle.OracleDB(g["prefix"])
    b = g["batch"]
    rc_flag, flag, T_off, T = db.scan_se(b)
    res = db.align_se(b, rc_flag, flag, T_off, T)
    tlen = formats.read_lengths(g["prefix"])
    cc = oracle.conclave(res["n_hits"], res["best_score"], b.length, np.zeros(b.n, np.int32), T_off[:-1], res["tmpl"],
                         res["start"], res["end"], res["alignment_scores"], res["uniq_alignment_scores"], tlen)
    st = oracle.res_stats(cc["w_scores"], tlen)
    return res, cc, st, tlen


def test_oracle_conclave_matches_res_and_frags(golden_se):
    """Stage 3b restatement vs the reference's own `.res` and `.frag.gz` (full pipeline run, tests/golden/se)."""
    res, cc, st, tlen = _oracle_conclave_se(golden_se)
    hdrs = [r["hdr"].rstrip(b"\0").decode() for r in golden_se["s1"]]
    seen, rows = golden_util.check_conclave_against_outputs("se", hdrs, res["n_hits"], cc["tmpl"], cc["w_scores"], st, tlen)
    assert seen > 900 and rows > 50


def test_oracle_conclave_in_mem_mode_matches_res(golden_se):
    """`-mem_mode` (runKMA_MEM, runkma.c:1090-1134): a stage-2 record is the frag_raw record -- its templates as the hits, each from 0
    to the template's length, the k-mer score as the read score, added to the ConClave vectors by update_Scores_MEM (updatescores.c:
    31-67) -- then ConClave and the `.res` statistics as ever: Score, Expected, Template_length, q_value and p_value of every row of the
    reference's own file (tests/golden/se/mem.res, make_golden_mem.py), and the template and tie count of every `.frag.gz` row"""
    import gzip
    g = golden_se
    db = oracle.OracleDB(g["prefix"])
    b = g["batch"]
    rc_flag, flag, T_off, T = db.scan_se(b)
    tlen = formats.read_lengths(g["prefix"])
    n = b.n
    n_hits = np.zeros(n, np.int32); score = np.abs(rc_flag).astype(np.int32)
    AS = np.zeros(len(tlen), np.uint64); US = np.zeros(len(tlen), np.uint64)
    for r in range(n):
        lst = T[T_off[r]:T_off[r + 1]]
        if len(lst) == 0 or b.length[r] < 16:
            score[r] = 0
            continue
        n_hits[r] = len(lst)
        for t in lst:
            AS[abs(int(t))] += np.uint64(score[r])
        if len(lst) == 1:
            US[abs(int(lst[0]))] += np.uint64(score[r])
    start = np.zeros(len(T), np.int32)
    end = tlen[np.abs(T)].astype(np.int32) if len(T) else np.zeros(0, np.int32)
    cc = oracle.conclave(n_hits, score, b.length, np.zeros(n, np.int32), T_off[:-1], T, start, end, AS, US, tlen)
    st = oracle.res_stats(cc["w_scores"], tlen)
    names = golden_util.template_names("se")
    want = {}
    for line in open(os.path.join(g["dir"], "mem.res")):
        if not line.startswith("#"):
            c = [x.strip() for x in line.rstrip("\n").split("\t")]
            want[c[0]] = (int(c[1]), int(c[2]), int(c[3]), c[9], c[10])
    assert len(want) > 50 and want != golden_util.load_res("se")
    rows = 0
    for t in range(1, len(tlen)):
        if cc["w_scores"][t] and names[t - 1] in want:
            rows += 1
            got = (int(cc["w_scores"][t]), int(st["expected"][t]), int(tlen[t]), "%.2f" % st["q_value"][t], "%4.1e" % st["p_value"][t])
            assert got == want[names[t - 1]], (names[t - 1], got, want[names[t - 1]])
    assert rows == len(want)
    hdrs = {r["hdr"].rstrip(b"\0").decode(): i for i, r in enumerate(g["s1"])}
    seen = 0
    for line in gzip.open(os.path.join(g["dir"], "mem.frag.gz"), "rt"):
        c = line.rstrip("\n").split("\t")
        i = hdrs[c[6]]
        seen += 1
        assert int(c[1]) == n_hits[i] and c[5] == names[abs(int(cc["tmpl"][i])) - 1], (c[6], c[1], c[5])
    assert seen > 900


def test_oracle_conclave_matches_res_and_frags_long_reads(golden_long):
    res, cc, st, tlen = _oracle_conclave_se(golden_long)
    hdrs = [r["hdr"].rstrip(b"\0").decode() for r in golden_long["s1"]]
    seen, rows = golden_util.check_conclave_against_outputs("long", hdrs, res["n_hits"], cc["tmpl"], cc["w_scores"], st, tlen)
    assert rows > 0      # this fixture holds the `.res` only


def test_oracle_conclave_matches_res_paired(golden_pe):
    """`-ipe ... -apm p`: a proper pair is ONE ConClave record whose score counts once (conclave.c:147, 171-175)."""
    import pe_util
    r = pe_util.oracle_pe_conclave_records(golden_pe)
    tlen = formats.read_lengths(golden_pe["prefix"])
    cc = oracle.conclave(r["n_hits"], r["score"], r["q_len"], r["q_len2"], r["off"], r["tmpl"], r["start"], r["end"],
                         r["alignment_scores"], r["uniq_alignment_scores"], tlen)
    st = oracle.res_stats(cc["w_scores"], tlen)
    seen, rows = golden_util.check_conclave_against_outputs("pe", [], r["n_hits"], cc["tmpl"], cc["w_scores"], st, tlen)
    assert rows > 50


def _oracle_trace_case(g, name):
    """Stage 3c per read: every read ConClave filed under a template, re-aligned with the traceback aligner, must give the
    POS / CIGAR / AS / MAPQ / FLAG of the reference's SAM record; reads the reference dropped in 3c must be dropped."""
    from kma_amd import synth
    res, cc, st, tlen = _oracle_conclave_se(g)
    sam = golden_util.load_sam(name)
    names = golden_util.template_names(name)
    al = oracle.OracleAligner(oracle.OracleDB(g["prefix"]))
    seen = 0
    for i, r in enumerate(g["s1"]):
        h = r["hdr"].rstrip(b"\0").decode()
        tt = int(cc["tmpl"][i])
        if tt == 0 or not st["significant"][abs(tt)]:
            assert h not in sam, h
            continue
        read = g["reads"][i]
        flag = int(res["out_flag"][i])
        if flag & 16:
            read = synth.revcomp_codes(read)
        if tt < 0:
            read = synth.revcomp_codes(read)
            flag |= 16
        o = al.align_trace(read, abs(tt))
        if o is None:
            assert h not in sam, h
            continue
        seen += 1
        exp = sam[h][0]
        got = (flag, names[abs(tt) - 1], o["start"] + 1, min(254, o["mapQ"]), o["cigar"], o["score"])
        assert got == exp, (h, got, exp)
    assert seen == len(sam)
    return seen


def test_oracle_traceback_matches_reference_sam(golden_se):
    assert _oracle_trace_case(golden_se, "se") > 900


def test_oracle_traceback_matches_reference_sam_long_reads(golden_long):
    assert _oracle_trace_case(golden_long, "long") > 200


def _oracle_assembly_case(g, name):
    """Stage 3c per template: reads in the order the reference assembles them (reverse stream order: ConClave prepends to
    the per-template list, conclave.c:164-165), piled up and called; the five consensus columns of every `.res` row must
    come out as printed, and a template without a row must fail the row's own gate."""
    from kma_amd import synth
    res, cc, st, tlen = _oracle_conclave_se(g)
    names = golden_util.template_names(name)
    exp = golden_util.load_res_identity(name)
    odb = oracle.OracleDB(g["prefix"])
    al = oracle.OracleAligner(odb)
    per_t, fsa = {}, []
    for i in range(len(g["s1"]) - 1, -1, -1):
        tt = int(cc["tmpl"][i])
        if tt and st["significant"][abs(tt)]:
            per_t.setdefault(abs(tt), []).append(i)
    rows = 0
    for t in range(1, len(tlen)):
        if not (cc["w_scores"][t] > 0 and st["significant"][t]):
            assert names[t - 1] not in exp
            continue
        asm = oracle.Assembly(odb, t, tlen[t])
        for i in per_t.get(t, []):
            read = g["reads"][i]
            if int(res["out_flag"][i]) & 16:
                read = synth.revcomp_codes(read)
            if int(cc["tmpl"][i]) < 0:
                read = synth.revcomp_codes(read)
            o = al.align_trace(read, t)
            if o is not None:
                asm.add(o, read)
        call = asm.call() if asm.n else None
        got = oracle.res_identity_columns(call, int(tlen[t])) if asm.n else None
        assert got == exp.get(names[t - 1]), (names[t - 1], got, exp.get(names[t - 1]))
        rows += got is not None
        if got is not None:
            fsa.append((names[t - 1], call["consensus"]))
    assert rows == len(exp)
    assert golden_util.fsa_text(fsa) == golden_util.load_fsa(name)
    return rows


def test_oracle_assembly_matches_res_identity_columns(golden_se):
    assert _oracle_assembly_case(golden_se, "se") > 50


def test_oracle_assembly_matches_res_identity_columns_long_reads(golden_long):
    assert _oracle_assembly_case(golden_long, "long") > 0


# ---- BASELINE config C4 in small: `-Mt1 1 -bcNano` on ONT-like reads (tests/golden/mt1, make_golden_mt1.py) ----------------
def _oracle_mt1(tmp_path):
    """every raw read through the restated anker_rc + KMA() + read filter, in stream order"""
    g = golden_util.load_mt1(tmp_path / "mt1")
    odb = oracle.OracleDB(g["prefix"])
    al = oracle.OracleAligner(odb)
    out = [al.align_trace_mt1(rd, 1) for rd in g["reads"]]
    return g, odb, out


def test_oracle_mt1_traceback_matches_reference_sam(tmp_path):
    """POS / CIGAR / AS / MAPQ of every read against the reference's SAM records: strand choice by MEM coverage, preseed,
    chain over hundreds of MEMs, full and banded joins with traceback. The flag stays 0 on the reverse strand in this mode
    (anker_rc leaves it alone, align.c:968-975), the sequence is what tells."""
    g, odb, out = _oracle_mt1(tmp_path)
    sam = golden_util.load_sam("mt1")
    mapped = 0
    for nm, (o, is_rc, rd) in zip(g["names"], out):
        flag, rname, pos, mapq, cigar, AS = sam[nm][0]
        if o is None:
            assert cigar == "*" and flag == 4, nm
            continue
        mapped += 1
        assert (0, "genome60k", o["start"] + 1, min(254, o["mapQ"]), o["cigar"], o["score"]) == (flag, rname, pos, mapq, cigar, AS), nm
    assert mapped == 174 and sum(1 for o, r, _ in out if o is not None and r) > 50


def test_oracle_mt1_pileup_and_nanocaller_match_res_fsa_frags(tmp_path):
    """reads piled up in stream order (one thread reads the records one by one, assembly.c:1873-1965), consensus by nanoCaller
    with significantAnd90Nuc: the `.res` row, the consensus FASTA and the `.frag.gz` rows of the reference."""
    g, odb, out = _oracle_mt1(tmp_path)
    tlen = formats.read_lengths(g["prefix"])
    asm = oracle.Assembly(odb, 1, tlen[1])
    rows = []
    score = 0
    for nm, (o, is_rc, rd) in zip(g["names"], out):
        if o is None:
            continue
        asm.add(o, rd)
        # alnToMat sums KMA()'s own score (assembly.c:1328-1334), i.e. without the end bonus Wl the read filter added
        score += o["score"] - odb.rw.Wl * ((o["start"] == 0) + (o["end"] == tlen[1]))
        rows.append(("".join("ACGTN"[c] for c in rd), "1", str(o["score"]), str(o["start"]), str(o["end"]), "genome60k", nm))
    assert rows == golden_util.load_frag_rows("mt1")
    call = asm.call(caller=1, sig=1)
    assert golden_util.fsa_text([("genome60k", call["consensus"])]) == golden_util.load_fsa("mt1")
    # the row as runKMA_Mt1 prints it (mt1.c:441-443): Score = summed read scores, Expected 0, q_value = score
    exp = open(os.path.join(golden_util.GOLD, "mt1", "out.res")).read().splitlines()[1].split("\t")
    assert int(exp[1]) == score and int(exp[3]) == tlen[1]
    assert tuple(x.strip() for x in exp[4:9]) == oracle.res_identity_columns(call, int(tlen[1]))


@pytest.mark.parametrize("name", ["se", "long"])
def test_chain_finder_oracle_matches_reference_s2_tap(tmp_path, name):
    """KMA's default template finder (no -1t1), oracle/chain.c, against the reference's own `-s2` tap of the same reads
    (tests/golden/make_golden_chain.py): the same records in the same order -- score / strand flag, template list, the read or
    its reverse complement, and the query bounds behind the header. Left out: reads with an N among their first k - 1 bases --
    behind an N the reference restarts its reverse-strand k-mer k bases too far on (savekmers.c:5447-5449; restated as it is),
    and for those reads that lands beyond the end of its buffer, on whatever an earlier read left there."""
    import gzip
    import struct
    g = golden_util.load_se(tmp_path, name)
    tap, n_reads = formats.parse_s2(gzip.open(os.path.join(g["dir"], "s2_chain.bin.gz")).read())
    by_read = {}
    for w in tap:
        by_read.setdefault(w["hdr"][:len(w["hdr"]) - 9], []).append(w)
    db = oracle.OracleDB(g["prefix"])
    got = db.scan_chain(g["batch"])
    b = g["batch"]
    n_rec = skipped = 0
    for i, recs in enumerate(got):
        hdr = g["s1"][i]["hdr"]
        Ni = b.N[b.N_off[i]:b.N_off[i + 1]]
        if len(Ni) and int(Ni[0]) < 15:
            skipped += 1
            continue
        want = by_read.get(hdr, [])
        assert len(want) == len(recs), (i, hdr, len(want), len(recs))
        for w, (rc_flag, emit_rc, q_start, q_end, T) in zip(want, recs):
            n_rec += 1
            assert w["hdr"] == hdr + b"\x00" + struct.pack("<2i", q_start, q_end), (i, hdr, w["hdr"], q_start, q_end)
            assert w["rc_flag"] == rc_flag and np.array_equal(w["T"], T), (i, hdr, w["rc_flag"], rc_flag, w["T"], T)
            L = int(b.length[i])
            seq = b.seq[b.seq_off[i]:b.seq_off[i] + ((L + 31) >> 5)]
            if emit_rc:
                seq, _ = oracle.rc_packed(seq, L, Ni)
            assert np.array_equal(w["seq"], seq), (i, hdr, "sequence / strand")
    assert n_rec > 200 and skipped < 100, (n_rec, skipped)


def _chimeric_reads(seqs, n, rng, with_n=True):
    """reads glued from 1-3 pieces of different genes and strands, with substitutions, small indels and (behind base 20) N's"""
    from kma_amd import synth
    out = []
    for _ in range(n):
        parts = []
        for _p in range(int(rng.integers(1, 4))):
            s = seqs[int(rng.integers(0, len(seqs)))]
            L = int(rng.integers(40, 260))
            a = int(rng.integers(0, max(1, len(s) - L)))
            r = s[a:a + L].copy()
            if rng.random() < 0.5:
                r = synth.revcomp_codes(r)
            parts.append(r)
            if rng.random() < 0.3:
                parts.append(rng.integers(0, 4, int(rng.integers(5, 60)), dtype=np.uint8))       # foreign stretch
        r = np.concatenate(parts)
        x = rng.random(len(r)) < 0.01
        r[x] = (r[x] + rng.integers(1, 4, int(x.sum()), dtype=np.uint8)) & 3
        if rng.random() < 0.2 and len(r) > 60:
            p = int(rng.integers(30, len(r) - 20))
            r = np.concatenate([r[:p], r[p + int(rng.integers(1, 4)):]]) if rng.random() < 0.5 else np.concatenate([r[:p], rng.integers(0, 4, int(rng.integers(1, 4)), dtype=np.uint8), r[p:]])
        if with_n and rng.random() < 0.15 and len(r) > 50:
            r[int(rng.integers(20, len(r)))] = 4
        out.append(np.ascontiguousarray(r.astype(np.uint8)))
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_chain_finder_oracle_differential_against_reference_binary(tmp_path, seed):
    """Chimeric reads (several chains per read: the extraction loop, tie anchors, the segment tree) through oracle/chain.c and
    through the compiled reference (`kma -s2` without -1t1). Skipped where oracle/_ref/kma has not been built."""
    import struct
    import subprocess
    from kma_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kma = os.path.join(root, "oracle", "_ref", "kma")
    if not os.path.exists(kma):
        pytest.skip("oracle/_ref/kma not built")
    rng = np.random.default_rng(100 + seed)
    names, seqs = synth.make_gene_db(25, 4, 300, 900, 0.05, seed=200 + seed)
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([kma, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = _chimeric_reads(seqs, 3000, rng)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads)
    tap = subprocess.run([kma, "-i", fq, "-o", str(tmp_path / "o"), "-t_db", prefix, "-t", "1", "-s2"], check=True, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL).stdout
    want, _ = formats.parse_s2(tap)
    db = oracle.OracleDB(prefix)
    b = formats.pack_ragged(reads)
    got = db.scan_chain(b)
    flat = []
    for i, recs in enumerate(got):
        for rc_flag, emit_rc, q_start, q_end, T in recs:
            flat.append((b"r%d" % i + bytes(2) + struct.pack("<2i", q_start, q_end), rc_flag, tuple(int(x) for x in T)))
    ref = [(w["hdr"], w["rc_flag"], tuple(int(x) for x in w["T"])) for w in want]
    multi = sum(1 for recs in got if len(recs) > 1)
    assert multi > 300, multi
    for x, (a, c) in enumerate(zip(flat, ref)):
        assert a == c, (x, a, c)
    assert len(flat) == len(ref)
