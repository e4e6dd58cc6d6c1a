"""Stage 3c per read on the GPU: KMA() with traceback vs the oracle and vs the reference's SAM records."""
import numpy as np
import pytest

import golden_util
import oracle
from kma_amd import formats

pytestmark = pytest.mark.gpu


def _case(g, name):
    from kma_amd import binding, synth
    db = binding.KmaHipDB(g["prefix"])
    try:
        b = g["batch"]
        (rc_flag, flag, T_off, T), h = db.map_se(b)
        cc = db.conclave_se(b.length, T_off, h)
        rows = db.res_rows(cc["w_scores"])
        ok = np.zeros(int(db.info.DB_size), np.uint8)
        for r in rows:
            ok[r.template_id] = r.significant
        stats, off, nops, ops = db.align_trace(b, h["rc"], cc["tmpl"], ok)
    finally:
        db.close()
    sam = golden_util.load_sam(name)
    names = golden_util.template_names(name)
    al = oracle.OracleAligner(oracle.OracleDB(g["prefix"]))
    seen = 0
    for i, r in enumerate(g["s1"]):
        hd = r["hdr"].rstrip(b"\0").decode()
        tt = int(cc["tmpl"][i])
        st = stats[i]
        if tt == 0 or not ok[abs(tt)]:
            assert not st.any() and hd not in sam, hd
            continue
        # oracle on the same read
        read = g["reads"][i]
        fl = int(h["flag"][i])
        if fl & 16:
            read = synth.revcomp_codes(read)
        if tt < 0:
            read = synth.revcomp_codes(read)
            fl |= 16
        o = al.align_trace(read, abs(tt))
        if o is None:
            assert not st.any() and hd not in sam, hd
            continue
        cigar = binding.cigar_from_runs(ops[off[i]:off[i] + nops[i]], int(st[4]), int(st[5]))
        got = dict(score=int(st[0]), start=int(st[1]), end=int(st[2]), aln_len=int(st[3]), clip_start=int(st[4]), clip_end=int(st[5]),
                   match=int(st[6]), tGaps=int(st[7]), qGaps=int(st[8]), mapQ=int(st[9]), cigar=cigar)
        o.pop("cols")
        assert got == o, (hd, got, o)
        seen += 1
        assert (fl, names[abs(tt) - 1], got["start"] + 1, min(254, got["mapQ"]), cigar, got["score"]) == sam[hd][0], hd
    assert seen == len(sam)
    return seen


def test_trace_matches_reference_sam(golden_se):
    assert _case(golden_se, "se") > 900


def test_trace_matches_reference_sam_long_reads(golden_long):
    assert _case(golden_long, "long") > 200


def _res_case(g, name):
    """Stages 2, 3a, 3b, 3c through the C-ABI: the `.res` file rebuilt line by line must equal the reference's."""
    import os
    from kma_amd import binding
    db = binding.KmaHipDB(g["prefix"])
    try:
        b = g["batch"]
        (rc_flag, flag, T_off, T), h = db.map_se(b)
        cc = db.conclave_se(b.length, T_off, h)
        rows = db.res_rows(cc["w_scores"])
        ok = np.zeros(int(db.info.DB_size), np.uint8)
        for r in rows:
            ok[r.template_id] = r.significant
        traces = db.align_trace(b, h["rc"], cc["tmpl"], ok)
        asm = db.assemble(b, h["rc"], cc["tmpl"], traces, consensus=True)
        names = golden_util.template_names(name)
        fsa = []
        lines = ["#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n"]
        for r in rows:
            if not r.significant:
                continue
            t = r.template_id
            line = db.res_line(names[t - 1], r, asm["cover"][t], asm["aln_len"][t], asm["depth"][t])
            if line:
                lines.append(line)
                fsa.append((names[t - 1], asm["consensus"][t]))
        # `.frag.gz`: every row (read as aligned, ties, score, start, end, template, header) in the reference's order
        if os.path.exists(os.path.join(golden_util.GOLD, name, "out.frag.gz")):
            import gzip
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                hdrs = [r["hdr"].rstrip(b"\0") for r in g["s1"]]
                for fn in ("x.frag.gz", "x.frag"):
                    n_rows = db.frag_write(os.path.join(tmp, fn), b, h["rc"], cc["tmpl"], h["n_hits"], traces[0], hdrs)
                    opener = gzip.open if fn.endswith(".gz") else open
                    got = opener(os.path.join(tmp, fn), "rb").read()
                    assert got == gzip.open(os.path.join(golden_util.GOLD, name, "out.frag.gz"), "rb").read()
                    assert n_rows == got.count(b"\n") > 0
    finally:
        db.close()
    with open(os.path.join(golden_util.GOLD, name, "out.res")) as f:
        exp = f.read()
    assert "".join(lines) == exp
    assert golden_util.fsa_text(fsa) == golden_util.load_fsa(name)
    # the oracle's consensus lines (same columns, same order) for the templates it assembles
    return len(lines) - 1, asm


def test_res_file_matches_reference_byte_for_byte(golden_se):
    rows, asm = _res_case(golden_se, "se")
    assert rows > 50


def test_res_file_matches_reference_byte_for_byte_long_reads(golden_long):
    rows, asm = _res_case(golden_long, "long")
    assert rows > 0
    assert (asm["asm_len"] > 0).sum() >= rows


def test_res_file_matches_reference_paired(golden_pe):
    """`-ipe ... -apm p` end to end: pairs and singly emitted reads through stages 2 / 3a, their records merged in stream
    order for ConClave, every fragment traced and piled up in the reference's order -> `.res`, consensus FASTA and the SAM
    records (POS, CIGAR, AS, FLAG per fragment) of the reference run."""
    import collections
    import os
    import pe_util
    from kma_amd import binding
    g = golden_pe
    name = "pe"
    db = binding.KmaHipDB(g["prefix"])
    try:
        cc = pe_util.hip_pe_conclave(db, g)
        rows = db.res_rows(cc["w_scores"])
        ok = np.zeros(int(db.info.DB_size), np.uint8)
        for r in rows:
            ok[r.template_id] = r.significant
        # one batch of fragments in frag_raw stream order; the first fragment of a record carries the sign of the template
        # (runConClave reverse-complements only it, conclave.c:131-146)
        reads, flags, rcs, tmpls, hdrs = [], [], [], [], []
        for k, fr in enumerate(cc["frags"]):
            tt = int(cc["tmpl"][k])
            for x, (i, fl, rcv) in enumerate(fr):
                r = g["s1"][i]
                reads.append(pe_util.codes_of(r["seq"], r["seqlen"], r["N"]))
                flags.append(fl)
                rcs.append(rcv)
                tmpls.append(tt if x == 0 else abs(tt))
                hdrs.append(r["hdr"].rstrip(b"\0").decode())
        b = formats.pack_ragged(reads)
        flags, rcs, tmpls = np.array(flags, np.int32), np.array(rcs, np.int32), np.array(tmpls, np.int32)
        traces = db.align_trace(b, rcs, tmpls, ok)
        asm = db.assemble(b, rcs, tmpls, traces, consensus=True)
        names = golden_util.template_names(name)
        lines = ["#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n"]
        fsa = []
        for r in rows:
            if r.significant:
                t = r.template_id
                line = db.res_line(names[t - 1], r, asm["cover"][t], asm["aln_len"][t], asm["depth"][t])
                if line:
                    lines.append(line)
                    fsa.append((names[t - 1], asm["consensus"][t]))
    finally:
        db.close()
    # SAM records per fragment (a pair has two records under one name)
    stats, off, nops, ops = traces
    got = collections.Counter()
    for i in range(b.n):
        st = stats[i]
        if not st[3]:
            continue
        tt = int(tmpls[i])
        cigar = binding.cigar_from_runs(ops[off[i]:off[i] + nops[i]], int(st[4]), int(st[5]))
        fl = int(flags[i]) | (16 if tt < 0 else 0)
        got[(hdrs[i], fl, names[abs(tt) - 1], int(st[1]) + 1, min(254, int(st[9])), cigar, int(st[0]))] += 1
    exp = collections.Counter()
    for h, recs in golden_util.load_sam(name).items():
        for rec in recs:
            exp[(h,) + rec] += 1
    assert got == exp
    with open(os.path.join(golden_util.GOLD, name, "out.res")) as f:
        assert "".join(lines) == f.read()
    assert golden_util.fsa_text(fsa) == golden_util.load_fsa(name)


def test_c1_full_size_res_and_consensus_match_reference(tmp_path):
    """BASELINE config C1 at full size (100 000 x 150 bp reads, 500 genes): the whole path -- stage 2, 3a, ConClave, the
    traceback aligner, pile-up, consensus -- through the C-ABI must give the reference's `.res` and consensus FASTA byte for
    byte. Inputs are regenerated from the seeds of tests/golden/make_golden_c1.py."""
    import gzip
    import os
    import sys
    from kma_amd import binding
    sys.path.insert(0, os.path.join(golden_util.GOLD))
    import make_golden_c1
    prefix, names, reads = make_golden_c1.inputs(str(tmp_path))
    b = formats.pack_fixed(reads)
    db = binding.KmaHipDB(prefix)
    try:
        (rc_flag, flag, T_off, T), h = db.map_se(b)
        cc = db.conclave_se(b.length, T_off, h)
        rows = db.res_rows(cc["w_scores"])
        ok = np.zeros(int(db.info.DB_size), np.uint8)
        for r in rows:
            ok[r.template_id] = r.significant
        traces = db.align_trace(b, h["rc"], cc["tmpl"], ok)
        asm = db.assemble(b, h["rc"], cc["tmpl"], traces, consensus=True)
        lines = ["#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n"]
        fsa = []
        for r in rows:
            if r.significant:
                t = r.template_id
                line = db.res_line(names[t - 1], r, asm["cover"][t], asm["aln_len"][t], asm["depth"][t])
                if line:
                    lines.append(line)
                    fsa.append((names[t - 1], asm["consensus"][t]))
    finally:
        db.close()
    with open(os.path.join(golden_util.GOLD, "c1", "out.res")) as f:
        assert "".join(lines) == f.read()
    with gzip.open(os.path.join(golden_util.GOLD, "c1", "out.fsa.gz"), "rt") as f:
        assert golden_util.fsa_text(fsa) == f.read()
    assert len(lines) == 501


def test_c_host_program_reproduces_reference_res_file(golden_se, golden_long):
    """examples/kmahip_res.c (plain C99 over the C-ABI): the reference's S1 stream in, the reference's `.res` out."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    exe = os.path.join(root, "examples", "kmahip_res")
    for g in (golden_se, golden_long):
        s1 = gzip.open(os.path.join(g["dir"], "s1.bin.gz"), "rb").read()
        out = subprocess.run([exe, "-t_db", g["prefix"]], input=s1, stdout=subprocess.PIPE, check=True).stdout
        assert out == open(os.path.join(g["dir"], "out.res"), "rb").read(), g["dir"]
        # the same from the FASTQ file itself: stage 1 by kmahip_ingest_* instead of the reference's S1 stream
        out = subprocess.run([exe, "-t_db", g["prefix"], "-i", os.path.join(g["dir"], "reads.fq.gz")], stdout=subprocess.PIPE, check=True).stdout
        assert out == open(os.path.join(g["dir"], "out.res"), "rb").read(), g["dir"]


def _one_call_case(g, name):
    """kmahip_run_se (reads uploaded once, everything else on the device) must give what the stage-wise calls give: the
    reference's `.res`, consensus FASTA and `.frag.gz`."""
    import gzip
    import os
    import tempfile
    from kma_amd import binding
    db = binding.KmaHipDB(g["prefix"])
    try:
        o = db.run_se(g["batch"])
        names = golden_util.template_names(name)
        lines = ["#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n"]
        fsa = []
        for r in o["rows"]:
            if not r.significant:
                continue
            t = r.template_id
            line = db.res_line(names[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
            if line:
                lines.append(line)
                fsa.append((names[t - 1], o["consensus"][t]))
        with open(os.path.join(golden_util.GOLD, name, "out.res")) as f:
            assert "".join(lines) == f.read()
        assert golden_util.fsa_text(fsa) == golden_util.load_fsa(name)
        if os.path.exists(os.path.join(golden_util.GOLD, name, "out.frag.gz")):
            with tempfile.TemporaryDirectory() as tmp:
                hdrs = [r["hdr"].rstrip(b"\0") for r in g["s1"]]
                db.frag_write(os.path.join(tmp, "x.frag.gz"), g["batch"], o["rc"], o["tmpl"], o["n_hits"], o["trace_stats"], hdrs)
                assert gzip.open(os.path.join(tmp, "x.frag.gz"), "rb").read() == gzip.open(os.path.join(golden_util.GOLD, name, "out.frag.gz"), "rb").read()
        assert len(o["ms"]) == 6 and all(m >= 0 for m in o["ms"])
        return len(lines) - 1
    finally:
        db.close()


def test_one_call_pipeline_matches_reference_files(golden_se, golden_long):
    assert _one_call_case(golden_se, "se") > 50
    assert _one_call_case(golden_long, "long") > 0


def test_device_consensus_equals_host_consensus(golden_se, golden_long, monkeypatch):
    """callConsensus on the device (significance as a threshold on the IEEE quotient) against the host arithmetic
    (KMAHIP_HOST_CONSENSUS=1: libm erf / tgamma per column) on the same pile-up: same figures, same consensus lines."""
    from kma_amd import binding
    for g in (golden_se, golden_long):
        res = []
        for host in (False, True):
            if host:
                monkeypatch.setenv("KMAHIP_HOST_CONSENSUS", "1")
            else:
                monkeypatch.delenv("KMAHIP_HOST_CONSENSUS", raising=False)
            db = binding.KmaHipDB(g["prefix"])
            try:
                o = db.run_se(g["batch"], per_read=False)
            finally:
                db.close()
            res.append(o)
        for key in ("cover", "aln_len", "depth", "asm_len"):
            assert np.array_equal(res[0][key], res[1][key]), key
        assert res[0]["consensus"] == res[1]["consensus"]
        assert (res[0]["asm_len"] > 0).sum() > 0


def test_c_program_from_fastq_to_all_three_output_files(golden_se, tmp_path):
    """examples/kmahip_map.c: reads.fq.gz + index in, out.res / out.fsa / out.frag.gz out -- the reference's files, with nothing of
    the reference in between (kmahip_ingest_*, kmahip_run_se, kmahip_res_line, kmahip_frag_write)."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    g = golden_se
    out = str(tmp_path / "out")
    subprocess.run([os.path.join(root, "examples", "kmahip_map"), "-i", os.path.join(g["dir"], "reads.fq.gz"), "-t_db", g["prefix"], "-o", out, "-1t1"],
                   check=True, stderr=subprocess.DEVNULL)
    assert open(out + ".res", "rb").read() == open(os.path.join(g["dir"], "out.res"), "rb").read()
    assert open(out + ".fsa").read() == golden_util.load_fsa("se")
    assert gzip.open(out + ".frag.gz", "rb").read() == gzip.open(os.path.join(g["dir"], "out.frag.gz"), "rb").read()


def test_host_program_in_mem_mode_matches_the_committed_reference_files(golden_se, tmp_path):
    """`-1t1 -mem_mode` on the committed single-end fixture against tests/golden/se/mem.* (the compiled reference's files, written by
    tests/golden/make_golden_mem.py): ConClave on the template finder's scores (kmahip_set_mem_mode), then stage 3c as ever"""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    g = golden_se
    for env in ({"KMAHIP_MAP_ONE_BATCH": "1"}, {"KMAHIP_MAP_BATCH": "300"}):
        out = str(tmp_path / ("out_" + next(iter(env))))
        subprocess.run([os.path.join(root, "examples", "kmahip_map"), "-i", os.path.join(g["dir"], "reads.fq.gz"), "-t_db", g["prefix"], "-o", out, "-1t1", "-mem_mode"],
                       check=True, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
        assert open(out + ".res", "rb").read() == open(os.path.join(g["dir"], "mem.res"), "rb").read()
        assert open(out + ".fsa", "rb").read() == gzip.open(os.path.join(g["dir"], "mem.fsa.gz")).read()
        assert gzip.open(out + ".frag.gz", "rb").read() == gzip.open(os.path.join(g["dir"], "mem.frag.gz"), "rb").read()


def test_one_call_paired_run_matches_reference_files(golden_pe, tmp_path):
    """kmahip_run_pe on the two mate files as kmahip_ingest_* reads them: the `.res` and consensus FASTA of `kma -ipe r1 r2 -apm p -1t1`."""
    import gzip
    import os
    from kma_amd import binding
    src = os.path.join(golden_util.GOLD, "pe")
    with binding.Ingest(os.path.join(src, "r1.fq.gz"), os.path.join(src, "r2.fq.gz")) as ing:
        batch, names, pair = ing.next(1 << 30)
    db = binding.KmaHipDB(golden_pe["prefix"])
    try:
        o = db.run_pe(batch, names, pair, frag_path=str(tmp_path / "x.frag.gz"))
        tn = golden_util.template_names("pe")
        lines = ["#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n"]
        fsa = []
        for r in o["rows"]:
            if r.significant:
                t = r.template_id
                line = db.res_line(tn[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
                if line:
                    lines.append(line)
                    fsa.append((tn[t - 1], o["consensus"][t]))
    finally:
        db.close()
    with open(os.path.join(src, "out.res")) as f:
        assert "".join(lines) == f.read()
    assert golden_util.fsa_text(fsa) == golden_util.load_fsa("pe")
    rows = gzip.open(tmp_path / "x.frag.gz", "rt").read().splitlines()
    sam = golden_util.load_sam("pe")
    assert len(rows) == sum(len(v) for v in sam.values()) > 1000          # one row per SAM record of the reference run

def test_paired_fixture_in_the_default_mode_matches_the_committed_reference_files(golden_pe, tmp_path):
    """`kma -ipe r1 r2` without -1t1 on the committed paired fixture (tests/golden/pe/out_default.*, written by
    tests/golden/make_golden_pe_default.py from the compiled reference): couples by union pairing, the nineteen records that lost their
    mate through the chain finder (kmahip_ws_set_pe_chain) -- through the C host program and through the C-ABI from Python."""
    import gzip
    import os
    import subprocess
    from kma_amd import binding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(golden_util.GOLD, "pe")
    want_res = open(os.path.join(src, "out_default.res")).read()
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    out = str(tmp_path / "out")
    subprocess.run([os.path.join(root, "examples", "kmahip_map"), "-ipe", os.path.join(src, "r1.fq.gz"), os.path.join(src, "r2.fq.gz"), "-t_db", golden_pe["prefix"], "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    assert open(out + ".res").read() == want_res
    assert open(out + ".fsa", "rb").read() == gzip.open(os.path.join(src, "out_default.fsa.gz")).read()
    assert gzip.open(out + ".frag.gz").read() == gzip.open(os.path.join(src, "out_default.frag.gz")).read()
    with binding.Ingest(os.path.join(src, "r1.fq.gz"), os.path.join(src, "r2.fq.gz")) as ing:
        batch, names, pair = ing.next(1 << 30)
    db = binding.KmaHipDB(golden_pe["prefix"])
    try:
        db.params.apm = 1                                    # (no -apm: union pairing, kma.c:206)
        db.set_pe_chain()
        o = db.run_pe(batch, names, pair, frag_path=str(tmp_path / "x.frag.gz"))
        tn = golden_util.template_names("pe")
        lines = [want_res.splitlines(True)[0]]
        for r in o["rows"]:
            if r.significant:
                t = r.template_id
                line = db.res_line(tn[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
                if line:
                    lines.append(line)
        assert "".join(lines) == want_res
        assert gzip.open(tmp_path / "x.frag.gz").read() == gzip.open(os.path.join(src, "out_default.frag.gz")).read()
        db.set_pe_chain(False)                               # ... and back: the -1t1 files again
        db.params.apm = 0
        o = db.run_pe(batch, names, pair)
        lines = [want_res.splitlines(True)[0]]
        for r in o["rows"]:
            if r.significant:
                t = r.template_id
                line = db.res_line(tn[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
                if line:
                    lines.append(line)
        assert "".join(lines) == open(os.path.join(src, "out.res")).read()
    finally:
        db.close()


def test_pileup_in_lds_equals_pileup_in_hbm(golden_se, golden_long, monkeypatch):
    """The pile-up keeps a template's columns (and its insertion columns) in LDS when they fit; a template too long for it, or
    a launch whose insertion columns overflowed the LDS table (repeated on HBM), works on HBM. Same result either way:
    default; KMAHIP_PILE_NO_LDS=1 (HBM from the start); KMAHIP_PILE_LDS_NODES=1 (overflow after one insertion column)."""
    from kma_amd import binding
    for g in (golden_se, golden_long):
        res = []
        for env in ({}, {"KMAHIP_PILE_NO_LDS": "1"}, {"KMAHIP_PILE_LDS_NODES": "1"}):
            for k in ("KMAHIP_PILE_NO_LDS", "KMAHIP_PILE_LDS_NODES"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            db = binding.KmaHipDB(g["prefix"])
            try:
                res.append(db.run_se(g["batch"], per_read=False))
            finally:
                db.close()
        for o in res[1:]:
            for key in ("cover", "aln_len", "depth", "asm_len"):
                assert np.array_equal(res[0][key], o[key]), key
            assert res[0]["consensus"] == o["consensus"]
