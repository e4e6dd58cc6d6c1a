"""The C host program over several ranks (examples/kmahip_map -gpus N: one process per rank, kmahip_run_se_sharded, the
communicator of comm.hip) against the same program on one rank: `.res` and `.fsa` byte for byte, `.frag.gz` after inflating.
On the one-GPU test box every rank uses device 0 and the exchanges are staged through shared memory (KMAHIP_COMM=shm); with two
devices visible the RCCL backend runs as well. Also: the command lines BASELINE.json spells, given to the reference and to
kmahip_map alike, and inputs that break off."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from kma_amd import formats, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
MAP = os.path.join(ROOT, "examples", "kmahip_map")


def _case(tmp_path, n=14000, gz=False):
    """a 200-gene database and a stream of reads with substitutions, insertions and deletions (their pile-up depends on the order
    of the reads), unmappable reads, ragged lengths and a few N's"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    names, seqs = synth.make_gene_db(n_families=40, variants=5, seed=77)
    reads, *_ = synth.make_reads(seqs, n, read_len=150, sub_rate=0.01, random_frac=0.02, seed=78)
    rag = [r for r in reads]
    rng = np.random.default_rng(5)
    for g in rng.integers(0, len(seqs), 60):
        if len(seqs[g]) > 260:
            rag += synth.make_long_reads(seqs[g], 40, read_len=250, sub=0.01, dele=0.012, ins=0.012, seed=int(g) + 1)
    rag = [rag[i] for i in rng.permutation(len(rag))]
    for i in rng.integers(0, len(rag), 40):
        rag[i] = rag[i].copy()
        rag[i][int(rng.integers(0, len(rag[i])))] = 4
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, rag, prefix="q")
    if gz:
        subprocess.check_call(["gzip", "-1", fq])
        fq += ".gz"
    return prefix, fq


def _run(args, env=None, ok=True):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    r = subprocess.run([MAP] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if ok:
        assert r.returncode == 0, r.stderr.decode()[-3000:]
    return r


def _same_files(a, b):
    assert open(a + ".res", "rb").read() == open(b + ".res", "rb").read() and open(a + ".res").read().count("\n") > 30
    assert open(a + ".fsa", "rb").read() == open(b + ".fsa", "rb").read()
    assert open(a + ".aln", "rb").read() == open(b + ".aln", "rb").read() and os.path.getsize(a + ".aln") > 1000
    assert gzip.open(a + ".frag.gz").read() == gzip.open(b + ".frag.gz").read()


@pytest.mark.parametrize("world,mf,gz", [(2, None, False), (3, 1500, False), (2, 1700, True)])
def test_ranks_of_the_c_host_program_write_the_single_rank_files(tmp_path, world, mf, gz):
    prefix, fq = _case(tmp_path, gz=gz)
    extra = ["-mf", str(mf)] if mf else []
    _run(["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one"), "-1t1"] + extra)
    _run(["-gpus", str(world), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "many"), "-1t1"] + extra, env={"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})
    _same_files(str(tmp_path / "one"), str(tmp_path / "many"))
    assert not [f for f in os.listdir(tmp_path) if f.startswith("many.part")]      # the parts are gone


def test_ranks_over_rccl_when_two_devices_are_visible(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (the placement-independent logic is covered by the shm backend above)")
    prefix, fq = _case(tmp_path)
    _run(["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one"), "-1t1"])
    _run(["-gpus", "2", "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "many"), "-1t1"])
    _same_files(str(tmp_path / "one"), str(tmp_path / "many"))


def test_byte_ranges_of_the_readers_tile_the_file(tmp_path):
    """kmahip_ingest_open_part: the parts of a plain FASTQ file, read one after the other, are the whole file's records in order"""
    import ctypes as C
    from kma_amd import binding
    prefix, fq = _case(tmp_path, n=3000)
    L = binding.lib()
    L.kmahip_ingest_open_part.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]

    def part(p, parts):
        h, whole = C.c_void_p(), C.c_int(-1)
        assert L.kmahip_ingest_open_part(fq.encode(), None, None, p, parts, C.byref(h), C.byref(whole)) == 0
        assert whole.value == 0
        b = binding.ReadBatchC()
        assert L.kmahip_ingest_next(h, 1 << 62, C.byref(b)) == 0
        n = b.reads.n_reads
        off = np.ctypeslib.as_array(C.cast(b.name_off, C.POINTER(C.c_int64)), shape=(n + 1,)).copy() if n else np.zeros(1, np.int64)
        names = C.string_at(b.names, int(off[-1])).split(b"\0")[:n] if n else []
        lens = np.ctypeslib.as_array(C.cast(b.reads.len, C.POINTER(C.c_int32)), shape=(n,)).copy() if n else np.zeros(0, np.int32)
        L.kmahip_ingest_close(h)
        return names, lens
    all_names, all_lens = part(0, 1)
    assert len(all_names) > 3000
    for parts in (2, 3, 7):
        got = [part(p, parts) for p in range(parts)]
        assert [x for g in got for x in g[0]] == all_names
        assert np.array_equal(np.concatenate([g[1] for g in got]), all_lens)
        assert all(len(g[0]) > 0 for g in got)


def test_baseline_command_lines_give_the_reference_files(tmp_path):
    """the literal command lines of BASELINE.json's configs (scaled down), with the thread and output switches scripts pass"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    prefix, fq = _case(tmp_path, n=5000)
    names, seqs = synth.make_gene_db(n_families=40, variants=5, seed=77)
    m1, m2, _ = synth.make_pairs(seqs, 2000, seed=9)
    synth.write_fastq(str(tmp_path / "r1.fq"), list(m1), prefix="p")
    synth.write_fastq(str(tmp_path / "r2.fq"), list(m2), prefix="p")
    r1, r2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    cases = [
        ["-i", fq, "-t_db", prefix, "-1t1"],                                        # C1 / C2
        ["-i", fq, "-t_db", prefix, "-1t1", "-t", "4", "-nc", "-na", "-nf", "-mmap", "-status", "-verbose", "2", "-tmp", str(tmp_path) + "/"],   # the output switches of BASELINE.md R2 (+ switches without an effect on the files)
        ["-ipe", r1, r2, "-t_db", prefix, "-apm", "p", "-1t1", "-t", "1"],         # C3
        ["-ipe", r1, r2, "-t_db", prefix, "-1t1", "-t", "1"],                      # ... and without -apm: the union pairing (kma.c:206)
        ["-i", fq, "-t_db", prefix, "-1t1", "-mp", "30", "-ml", "40", "-eq", "25", "-mf", "900"],
        ["-i", fq, "-t_db", prefix, "-Mt1", "3", "-bcNano"],                       # C4's switches
        ["-i", fq, "-t_db", prefix, "-t", "2"],                                    # the default mode
        ["-ipe", r1, r2, "-t_db", prefix, "-t", "1"],                              # ... with paired input (single records through the chain finder)
        ["-i", fq, r1, r2, "-t_db", prefix, "-1t1", "-t", "1"],                    # lists of files (kma.c:371-435), read one after the other
        ["-i", fq, "-t_db", prefix, "-1t1", "-and"],                               # p-value AND score decide a template (kma.c:915: two thirds of the rows go)
        ["-i", fq, "-t_db", prefix, "-1t1", "-oa", "-5p", "3", "-3p", "2"],        # ... neither does; -5p / -3p are read and never used (runinput.c:127)
        ["-i", fq, "-t_db", prefix, "-Mt1", "3", "-and"],
        ["-ipe", r1, r2, r2, r1, "-t_db", prefix, "-apm", "p", "-1t1", "-t", "1"],
        ["-ipe", r1, r2, "-t_db", prefix, "-pm", "p", "-1t1", "-t", "1"],         # the stages' pairing set apart (kma.c:437-465): penalty in stage 2, union in 3a
        ["-ipe", r1, r2, "-t_db", prefix, "-apm", "p", "-pm", "u", "-1t1", "-t", "1"],
        ["-ipe", r1, r2, "-t_db", prefix, "-fpm", "p", "-t", "1"],
    ]
    for i, args in enumerate(cases):
        ref, got = str(tmp_path / f"ref{i}"), str(tmp_path / f"got{i}")
        subprocess.run([KMA] + args + ["-o", ref] + ([] if "-t" in args else ["-t", "1"]), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _run(args + ["-o", got])
        assert open(got + ".res", "rb").read() == open(ref + ".res", "rb").read(), args
        for ext, opener in ((".fsa", open), (".aln", open), (".frag.gz", gzip.open)):
            assert os.path.exists(got + ext) == os.path.exists(ref + ext), (args, ext)
            if os.path.exists(ref + ext):
                a, b = opener(got + ext, "rb").read(), opener(ref + ext, "rb").read()
                if ext == ".frag.gz" and "-t" in args and args[args.index("-t") + 1] != "1":
                    a, b = sorted(a.splitlines()), sorted(b.splitlines())      # (the reference's row order depends on its threads' timing)
                assert a == b, (args, ext)
    # what is not built is refused, not ignored
    for bad in (["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "x"), "-1t1", "-apm", "f"], ["-ipe", r1, r2, "-t_db", prefix, "-o", str(tmp_path / "x"), "-apm", "f"],
                ["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "x"), "-1t1", "-sam"]):
        assert _run(bad, ok=False).returncode != 0


def test_input_that_breaks_off_ends_the_run_with_an_error(tmp_path):
    """a truncated .gz and a record that is no FASTQ: the reference exits non-zero; so must the one-batch host program (it asks
    the reader once more behind the batch)"""
    prefix, fq = _case(tmp_path, n=3000)
    raw = open(fq, "rb").read()
    z = gzip.compress(raw, 1)
    (tmp_path / "cut.fq.gz").write_bytes(z[: len(z) // 2])
    cut = raw.index(b"\n@q", len(raw) // 2) + 1                                   # between two records
    (tmp_path / "bad.fq").write_bytes(raw[:cut] + b"this is no record\n" + raw[cut:])
    for f in ("cut.fq.gz", "bad.fq"):
        r = _run(["-i", str(tmp_path / f), "-t_db", prefix, "-o", str(tmp_path / "x"), "-1t1"], ok=False)
        assert r.returncode != 0, f
        assert b"ingest" in r.stderr


def _pe_case(tmp_path, n_pairs=12000, chimeras=False):
    rng = np.random.default_rng(5)
    names, seqs = synth.make_gene_db(30, 5, 700, 1400, 0.04, seed=77)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    m1, m2, _ = synth.make_pairs(seqs, n_pairs, seed=9)
    r1, r2 = [r.copy() for r in m1], [r.copy() for r in m2]
    q1, q2 = [b"I" * 150] * len(r1), [b"I" * 150] * len(r2)
    for i in rng.choice(len(r1), n_pairs // 20, replace=False):                   # a foreign mate
        (r1 if rng.random() < 0.5 else r2)[i] = rng.integers(0, 4, 150, dtype=np.uint8)
    for x, i in enumerate(rng.choice(len(r1), n_pairs // 20, replace=False)):     # a mate that the quality trim shortens below -ml: a single record
        q = bytearray(b"I" * 150)
        q[10:] = b"#" * 140
        first = rng.random() < 0.5
        if first:
            q1[i] = bytes(q)
        else:
            q2[i] = bytes(q)
        if chimeras and x % 3 == 0:                                               # ... whose mate is made of two genes (the default mode files it in pieces)
            g1, g2 = (seqs[int(g)] for g in rng.choice(len(seqs), 2, replace=False))
            a1, a2 = int(rng.integers(0, len(g1) - 80)), int(rng.integers(0, len(g2) - 80))
            piece2 = g2[a2:a2 + 75] if x % 2 else (3 - g2[a2:a2 + 75])[::-1]
            (r2 if first else r1)[i] = np.concatenate([g1[a1:a1 + 75], piece2]).astype(np.uint8)
    for i in rng.choice(len(r1), n_pairs // 12, replace=False):                   # an insertion or a deletion in a mate (the pile-up order matters)
        r = r1 if rng.random() < 0.5 else r2
        a = int(rng.integers(30, 120))
        r[i] = np.concatenate([r[i][:a], rng.integers(0, 4, 2, dtype=np.uint8), r[i][a:148]]) if rng.random() < 0.5 else np.concatenate([r[i][:a], r[i][a + 2:], r[i][:2]])
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for path, rs, qs, tag in ((tmp_path / "r1.fq", r1, q1, b"/1"), (tmp_path / "r2.fq", r2, q2, b"/2")):
        with open(path, "wb") as f:
            for i, (r, q) in enumerate(zip(rs, qs)):
                f.write(b"@p%d" % i + tag + b"\n" + lut[r].tobytes() + b"\n+\n" + q + b"\n")
    return prefix, str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")


@pytest.mark.parametrize("world,mf", [(2, None), (3, 7), (2, 1001)])
def test_ranks_of_the_paired_run_write_the_single_rank_files(tmp_path, world, mf):
    """`-ipe r1 r2 -apm p -1t1` over 2 and 3 ranks: the chunks of -mf fragments close along the whole stream (with -mf 7 thousands of
    them, many closed by a couple that straddles the limit, and every shard boundary falls into an open chunk)"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    prefix, r1, r2 = _pe_case(tmp_path)
    extra = ["-mf", str(mf)] if mf else []
    env = {"KMAHIP_ROW_GRAIN": "700"} if mf else {}
    _run(["-ipe", r1, r2, "-apm", "p", "-t_db", prefix, "-o", str(tmp_path / "one"), "-1t1"] + extra, env=env)
    _run(["-gpus", str(world), "-ipe", r1, r2, "-apm", "p", "-t_db", prefix, "-o", str(tmp_path / "many"), "-1t1"] + extra,
         env=dict(env, KMAHIP_COMM="shm", KMAHIP_SHARE_GPU="1"))
    _same_files(str(tmp_path / "one"), str(tmp_path / "many"))


def test_paired_fixture_with_empty_hit_lists_over_three_ranks(tmp_path):
    """the committed paired fixture holds records whose hit list came out empty (they take the first hit of the record before them,
    DESIGN 3.3): cut into three shards the hand-over between shards is exercised, and the files must still be the reference's"""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    g = golden_util.load_pe(tmp_path)
    for f in ("r1.fq", "r2.fq"):
        with gzip.open(os.path.join(g["dir"], f + ".gz")) as a, open(tmp_path / f, "wb") as b:
            shutil.copyfileobj(a, b)
    args = ["-ipe", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-apm", "p", "-t_db", g["prefix"], "-1t1"]
    for world in (1, 3, 5):
        out = str(tmp_path / f"w{world}")
        _run((["-gpus", str(world)] if world > 1 else []) + args + ["-o", out], env={"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})
        assert open(out + ".res").read() == open(os.path.join(g["dir"], "out.res")).read(), world
        assert open(out + ".fsa").read() == gzip.open(os.path.join(g["dir"], "out.fsa.gz"), "rt").read(), world


@pytest.mark.parametrize("batch,chunk,mf", [(1000, 20000, None), (777, 3000000, 1500), (1 << 20, None, None)])
def test_batched_session_writes_the_one_batch_files(tmp_path, batch, chunk, mf):
    """the single-end -1t1 run batch by batch (kmahip_session_*: reads and headers kept in HBM, fragment rows ordered and formatted
    on the device, text back in chunks) against the same program taking the whole input as one batch (kmahip_run_se + the host's
    writer): tiny batches, tiny text chunks, chunks of -mf fragments that straddle batches"""
    prefix, fq = _case(tmp_path)
    extra = ["-mf", str(mf)] if mf else []
    _run(["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one"), "-1t1"] + extra, env={"KMAHIP_MAP_ONE_BATCH": "1"})
    env = {"KMAHIP_MAP_BATCH": str(batch)}
    if chunk:
        env["KMAHIP_FRAG_CHUNK"] = str(chunk)
    r = _run(["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "many"), "-1t1"] + extra, env=env)
    assert b"batches" in r.stderr
    _same_files(str(tmp_path / "one"), str(tmp_path / "many"))


@pytest.mark.parametrize("batch,mf", [(777, 7), (5000, 1001), (1 << 20, None)])
def test_batched_paired_session_writes_the_one_batch_files(tmp_path, batch, mf):
    """`-ipe r1 r2 -apm p -1t1` batch by batch (kmahip_session_set_pe: the batches' reads and headers wait in HBM, the host holds one
    batch at a time, the run's stages on everything when the input has ended) against the same program taking the input as one batch
    (kmahip_run_pe): couples, singly filed mates (a mate the trim removes), foreign mates, indels; batches of 777 records -- a record is a
    couple or a single read, so the batches' read counts are ragged -- and chunks of -mf fragments that straddle them"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    prefix, r1, r2 = _pe_case(tmp_path, n_pairs=9000)
    extra = ["-mf", str(mf)] if mf else []
    args = ["-ipe", r1, r2, "-apm", "p", "-t_db", prefix, "-1t1"] + extra
    _run(args + ["-o", str(tmp_path / "one")], env={"KMAHIP_MAP_ONE_BATCH": "1"})
    r = _run(args + ["-o", str(tmp_path / "many")], env={"KMAHIP_MAP_BATCH": str(batch), "KMAHIP_ROW_GRAIN": "700"})
    assert b"batches" in r.stderr
    _same_files(str(tmp_path / "one"), str(tmp_path / "many"))


@pytest.mark.parametrize("apm,interleaved", [(None, False), ("p", False), (None, True)])
def test_paired_input_in_the_default_mode_writes_the_reference_files(tmp_path, apm, interleaved):
    """`-ipe r1 r2` WITHOUT -1t1, the reference's default: couples go to save_kmers_pair as ever, a record that lost its mate to the
    trimming goes to kmerScan = save_kmers_chain (savekmers.c:196-200) and is filed in pieces with query bounds -- here a third of
    those reads are made of two genes (one piece forward, one reversed). One batch, batch by batch and over three ranks against
    the compiled reference: `.res`, `.fsa`, `.aln` byte for byte, `.frag.gz` after inflating. interleaved: the same couples as ONE
    file given with `-int` (run_input_INT, runinput.c:608-740; kmahip_ingest_open_interleaved)."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    prefix, r1, r2 = _pe_case(tmp_path, n_pairs=6000, chimeras=True)
    args = ["-ipe", r1, r2, "-t_db", prefix] + (["-apm", apm] if apm else [])
    if interleaved:
        with open(r1, "rb") as f1, open(r2, "rb") as f2, open(tmp_path / "ilv.fq", "wb") as o:
            l1, l2 = f1.read().split(b"\n"), f2.read().split(b"\n")
            for i in range(0, len(l1) - 1, 4):
                o.write(b"\n".join(l1[i:i + 4]) + b"\n" + b"\n".join(l2[i:i + 4]) + b"\n")
        args = ["-int", str(tmp_path / "ilv.fq"), "-t_db", prefix]
    ref = str(tmp_path / "ref")
    subprocess.run([KMA] + args + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    one2one = str(tmp_path / "ref1")
    subprocess.run([KMA] + args + ["-o", one2one, "-t", "1", "-1t1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert gzip.open(ref + ".frag.gz").read() != gzip.open(one2one + ".frag.gz").read()          # (the case tells the two modes apart)
    _run(args + ["-o", str(tmp_path / "one")], env={"KMAHIP_MAP_ONE_BATCH": "1"})
    _same_files(ref, str(tmp_path / "one"))
    r = _run(args + ["-o", str(tmp_path / "many")], env={"KMAHIP_MAP_BATCH": "777", "KMAHIP_ROW_GRAIN": "700"})
    assert b"batches" in r.stderr
    _same_files(ref, str(tmp_path / "many"))
    _run(["-gpus", "3"] + args + ["-o", str(tmp_path / "ranks")], env={"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})
    _same_files(ref, str(tmp_path / "ranks"))


def test_pairing_of_the_two_stages_set_apart(tmp_path):
    """-pm x sets save_kmers_pair alone, -fpm x alnFragsPE alone (kma.c:437-465; kmahip_params.apm bits 0-1 / 4-5): couples, foreign
    mates, singly filed mates and indels through the penalty pairing in one stage and the union in the other -- the reference's
    four combinations give four different `.res` files on this input, and ours must be those"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    prefix, r1, r2 = _pe_case(tmp_path, n_pairs=3000, chimeras=True)
    seen = set()
    for i, opts in enumerate((["-pm", "p"], ["-fpm", "p"], ["-apm", "p", "-pm", "u"], ["-pm", "u", "-fpm", "u"])):
        args = ["-ipe", r1, r2, "-t_db", prefix, "-1t1"] + opts
        ref, got = str(tmp_path / f"ref{i}"), str(tmp_path / f"got{i}")
        subprocess.run([KMA] + args + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _run(args + ["-o", got])
        _same_files(ref, got)
        seen.add(open(ref + ".res", "rb").read())
    assert len(seen) >= 3


def test_lc_with_1t1_equals_the_reference(tmp_path):
    """-lc outside the chain finder = runConClave_lc (conclave.c:215-385): among a read's tied templates the ConClave score per
    template base decides before the score itself. Genes with a half-length copy as a template of its own and twice the reads on
    that half: without -lc no read goes to a half, with -lc the halves take theirs (twenty more rows in the `.res`)."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    names, seqs = synth.make_gene_db(20, 3, 600, 1200, 0.05, seed=5)
    names, seqs = list(names), list(seqs)
    rng = np.random.default_rng(3)
    n_genes = len(seqs)
    for g in range(0, n_genes, 3):
        names.append(names[g] + "_half")
        seqs.append(seqs[g][:len(seqs[g]) // 2].copy())
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads = []
    for g in range(n_genes):
        L = len(seqs[g])
        for _ in range(L // 30):
            a = int(rng.integers(0, L - 150))
            reads.append(seqs[g][a:a + 150].copy())
        if g % 3 == 0:
            for _ in range(L // 30):
                a = int(rng.integers(0, L // 2 - 150))
                reads.append(seqs[g][a:a + 150].copy())
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, [reads[i] for i in rng.permutation(len(reads))], prefix="q")
    res = {}
    for tag, opts in (("plain", []), ("lc", ["-lc"])):
        args = ["-i", fq, "-t_db", prefix, "-1t1"] + opts
        ref, got = str(tmp_path / ("ref" + tag)), str(tmp_path / ("got" + tag))
        subprocess.run([KMA] + args + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _run(args + ["-o", got])
        for ext, opener in ((".res", open), (".fsa", open), (".aln", open), (".frag.gz", gzip.open)):
            assert opener(got + ext, "rb").read() == opener(ref + ext, "rb").read(), (tag, ext)
        res[tag] = open(ref + ".res").read()
    assert res["plain"].count("_half") == 0 and res["lc"].count("_half") >= 10
    assert _run(["-i", fq, "-t_db", prefix, "-lc", "-o", str(tmp_path / "x")], ok=False).returncode != 0      # (the chain finder's -lc is not built: refused)


@pytest.mark.parametrize("mode", ["1t1", "default", "pe_p", "pe_default"])
def test_mem_mode_equals_the_reference(tmp_path, mode):
    """-mem_mode (runKMA_MEM, runkma.c:910-1250): ConClave on the template finder's own scores, no alignment before it -- the `.res`
    scores are sums of k-mer scores and differ from the plain run's. -1t1 on reads with indels and N's, the default mode on reads that
    map in pieces; one batch, batch by batch and over three ranks against the compiled reference."""
    import sys
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    if mode == "1t1":
        prefix, fq = _case(tmp_path, n=9000)
        args = ["-i", fq, "-t_db", prefix, "-1t1", "-mem_mode"]
    elif mode.startswith("pe"):          # couples (one record with both scores, update_Scores_pe_MEM), foreign mates, mates filed singly -- and, in the default mode, in pieces
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
        prefix, r1, r2 = _pe_case(tmp_path, n_pairs=5000, chimeras=True)
        args = ["-ipe", r1, r2, "-t_db", prefix] + (["-apm", "p", "-1t1"] if mode == "pe_p" else []) + ["-mem_mode"]
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_oracle_golden import _chimeric_reads
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
        rng = np.random.default_rng(77)
        names, seqs = synth.make_gene_db(40, 5, 300, 900, 0.05, seed=913)
        prefix = str(tmp_path / "db")
        formats.write_index(prefix, names, seqs)
        fq = str(tmp_path / "r.fq")
        synth.write_fastq(fq, _chimeric_reads(seqs, 7000, rng, with_n=False))
        args = ["-i", fq, "-t_db", prefix, "-mem_mode"]
    ref, plain = str(tmp_path / "ref"), str(tmp_path / "plain")
    subprocess.run([KMA] + args + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run([KMA] + args[:-1] + ["-o", plain, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert open(ref + ".res").read() != open(plain + ".res").read()
    for name, pre, env in (("one", [], {"KMAHIP_MAP_ONE_BATCH": "1"}), ("many", [], {"KMAHIP_MAP_BATCH": "777"}),
                           ("ranks", ["-gpus", "3"], {"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})):
        got = str(tmp_path / name)
        _run(pre + args + ["-o", got], env=env)
        for ext, opener in ((".res", open), (".fsa", open), (".aln", open), (".frag.gz", gzip.open)):
            assert opener(got + ext, "rb").read() == opener(ref + ext, "rb").read(), (name, ext)


def test_mt1_with_paired_input_equals_the_reference(tmp_path):
    """`-Mt1 n -ipe r1 r2` (printFsa_pairMt1, mt1.c:61-83): the mates of a couple are records of their own, the second one reverse
    complemented; a mate that lost its partner to the trimming is a record as it is. One batch, batch by batch, three ranks."""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    prefix, r1, r2 = _pe_case(tmp_path, n_pairs=4000, chimeras=True)
    args = ["-ipe", r1, r2, "-t_db", prefix, "-Mt1", "12", "-bcNano"]
    ref = str(tmp_path / "ref")
    subprocess.run([KMA] + args + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert gzip.open(ref + ".frag.gz").read().count(b"\n") > 100
    for name, pre, env in (("one", [], {"KMAHIP_MAP_ONE_BATCH": "1"}), ("many", [], {"KMAHIP_MAP_BATCH": "777"}),
                           ("ranks", ["-gpus", "3"], {"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})):
        got = str(tmp_path / name)
        _run(pre + args + ["-o", got], env=env)
        for ext, opener in ((".res", open), (".fsa", open), (".aln", open), (".frag.gz", gzip.open)):
            assert opener(got + ext, "rb").read() == opener(ref + ext, "rb").read(), (name, ext)


SCHEMES = {
    "cge": ["-cge"],                                                                                  # kma.c:1024-1030 (what CGE's tools pass)
    "own": ["-reward", "2", "-penalty", "7", "-gapopen", "5", "-gapextend", "2", "-transition", "1", "-transversion", "4", "-localopen", "8",
            "-Npenalty", "1", "-per", "9"],
}


@pytest.mark.parametrize("scheme", sorted(SCHEMES))
def test_scoring_scheme_options_equal_the_reference(tmp_path, scheme):
    """-reward / -gapopen / -gapextend / -localopen / -Npenalty / -per / -transition / -transversion and the -cge preset (kma.c:821-915,
    1024-1030; -penalty is parsed and then replaced by the mean of transition and transversion, kma.c:1308): every stage reads its
    scores from kmahip_params.rw, so the files must be the reference's under another scheme too -- -1t1, the default mode on reads
    that map in pieces, couples with the pairing reward, and -Mt1."""
    import sys
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import _chimeric_reads
    opts = SCHEMES[scheme]
    (tmp_path / "se").mkdir(); (tmp_path / "pe").mkdir(); (tmp_path / "ch").mkdir()
    prefix, fq = _case(tmp_path / "se", n=6000)
    pprefix, r1, r2 = _pe_case(tmp_path / "pe", n_pairs=4000, chimeras=True)
    rng = np.random.default_rng(31)
    names, seqs = synth.make_gene_db(40, 5, 300, 900, 0.05, seed=912)
    cprefix = str(tmp_path / "ch" / "db")
    formats.write_index(cprefix, names, seqs)
    cfq = str(tmp_path / "ch" / "r.fq")
    synth.write_fastq(cfq, _chimeric_reads(seqs, 5000, rng, with_n=True))
    cases = [["-i", fq, "-t_db", prefix, "-1t1"], ["-i", cfq, "-t_db", cprefix], ["-ipe", r1, r2, "-t_db", pprefix, "-apm", "p", "-1t1"],
             ["-ipe", r1, r2, "-t_db", pprefix], ["-i", fq, "-t_db", prefix, "-Mt1", "7", "-bcNano"]]
    for i, args in enumerate(cases):
        ref, got = str(tmp_path / f"ref{i}"), str(tmp_path / f"got{i}")
        subprocess.run([KMA] + args + opts + ["-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        plain = str(tmp_path / f"plain{i}")
        subprocess.run([KMA] + args + ["-o", plain, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        if scheme == "own":
            assert gzip.open(ref + ".frag.gz").read() != gzip.open(plain + ".frag.gz").read(), args      # (the scheme shows in the files)
        _run(args + opts + ["-o", got])
        for ext, opener in ((".res", open), (".fsa", open), (".aln", open), (".frag.gz", gzip.open)):
            assert opener(got + ext, "rb").read() == opener(ref + ext, "rb").read(), (args, ext)


@pytest.mark.parametrize("world,gz,bc", [(2, False, True), (3, True, True), (2, False, False)])
def test_mt1_over_ranks_writes_the_single_rank_files_and_the_reference_s(tmp_path, world, gz, bc):
    """`-Mt1 1 [-bcNano]` (kmahip_run_mt1_sharded): every rank traces its part of the stream, the kept reads meet at rank 0 with
    their positions in the whole stream and are piled up in stream order -- long noisy reads, whose insertion columns depend on it"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(31)
    genome = rng.integers(0, 4, 60_000, dtype=np.uint8)
    prefix = str(tmp_path / "g")
    synth.write_fasta(prefix + ".fsa", ["genome"], [genome])
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_long_reads(genome, 400, read_len=3000, seed=5)
    reads += [rng.integers(0, 4, 800, dtype=np.uint8) for _ in range(7)]          # reads from elsewhere: not kept
    reads = [reads[i] for i in rng.permutation(len(reads))]
    fq = str(tmp_path / "ont.fq")
    synth.write_fastq(fq, reads, prefix="r", qual=b"5")
    if gz:
        subprocess.check_call(["gzip", "-1", fq])
        fq += ".gz"
    opt = ["-Mt1", "1"] + (["-bcNano"] if bc else [])
    _run(["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one")] + opt)
    _run(["-gpus", str(world), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "many")] + opt, env={"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})
    a, b = str(tmp_path / "one"), str(tmp_path / "many")
    assert open(a + ".res", "rb").read() == open(b + ".res", "rb").read() and open(a + ".res").read().count("\n") == 2
    assert open(a + ".fsa", "rb").read() == open(b + ".fsa", "rb").read()
    assert gzip.open(a + ".frag.gz").read() == gzip.open(b + ".frag.gz").read()
    assert not [f for f in os.listdir(tmp_path) if f.startswith("many.part")]
    if os.path.exists(KMA):
        subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"] + opt, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = str(tmp_path / "ref")
        assert open(r + ".res").read() == open(b + ".res").read()
        assert open(r + ".fsa").read() == open(b + ".fsa").read()
        assert gzip.open(r + ".frag.gz").read() == gzip.open(b + ".frag.gz").read()


@pytest.mark.parametrize("world,mf,seed,more", [(2, None, 1, []), (3, 777, 2, ["-mrc", "0.7"]), (5, 60, 3, [])])
def test_default_mode_over_ranks_writes_the_single_rank_files_and_the_reference_s(tmp_path, world, mf, seed, more):
    """no -1t1 (kmahip_run_chain_sharded): reads that map in pieces -- several records per read, strand ties, query bounds -- with N's
    and indels; the records of the shards must come out as the records of the one stream (their pile-up and fragment rows follow the
    order of the whole stream, the chunks of -mf filed fragments straddle the shards)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import _chimeric_reads
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(900 + seed)
    names, seqs = synth.make_gene_db(40, 5, 300, 900, 0.05, seed=910 + seed)
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = _chimeric_reads(seqs, 9000, rng, with_n=seed > 1)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads)
    extra = (["-mf", str(mf)] if mf else []) + more          # (-mrc: stage 2 is as without it, kmeranker.c:57-81; the pieces of a read fail the aligner's test)
    _run(["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one")] + extra)
    _run(["-gpus", str(world), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "many")] + extra, env={"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"})
    a, b = str(tmp_path / "one"), str(tmp_path / "many")
    assert open(a + ".res", "rb").read() == open(b + ".res", "rb").read() and open(a + ".res").read().count("\n") > 30
    assert open(a + ".fsa", "rb").read() == open(b + ".fsa", "rb").read()
    fa, fb = gzip.open(a + ".frag.gz").read(), gzip.open(b + ".frag.gz").read()
    assert fa == fb and fa.count(b"\n") > 3000
    assert not [f for f in os.listdir(tmp_path) if f.startswith("many.part")]
    subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"] + extra, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = str(tmp_path / "ref")
    assert open(r + ".res").read() == open(b + ".res").read()
    assert open(r + ".fsa").read() == open(b + ".fsa").read()
    assert gzip.open(r + ".frag.gz").read() == fb


@pytest.mark.parametrize("opts", [["-bc90"], ["-bcg"], ["-bc", "0.7"], ["-ref_fsa"], ["-ref_fsa", "0"], ["-bcNano", "-ref_fsa"], ["-bc", "0.6", "-bcNano"],
                                  ["-bcNano", "-bcg"], ["-bcd", "12", "-bc90"], ["-dense"], ["-dense", "-bcNano", "-ref_fsa", "0"]])
def test_base_caller_options_equal_the_reference(tmp_path, opts):
    """the consensus options of kma.c:671-770 (orgBaseCaller, refCaller, refNanoCaller, significantAnd90Nuc, significantAndSupport,
    the three forms of the consensus file) through the batched -1t1 run, the one-batch default mode and three ranks: `.res` and `.fsa`
    of the compiled reference with the same options"""
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    names, seqs = synth.make_gene_db(n_families=12, variants=4, seed=177)
    rng = np.random.default_rng(9)
    reads = []
    # shallow and deep templates, noisy reads with indels: calls below the depth and support thresholds, gaps that are and are not significant
    for g in range(len(seqs)):
        depth = int(rng.choice([2, 6, 25, 80]))
        reads += synth.make_long_reads(seqs[g], depth, read_len=min(220, len(seqs[g]) - 5), sub=0.04, dele=0.03, ins=0.03, seed=1000 + g)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    fq = str(tmp_path / "r.fq")
    synth.write_fastq(fq, reads)
    for mode, env, pre in ((["-1t1"], {}, []), ([], {"KMAHIP_MAP_ONE_BATCH": "1"}, []), (["-1t1"], {"KMAHIP_COMM": "shm", "KMAHIP_SHARE_GPU": "1"}, ["-gpus", "3"])):
        subprocess.run([KMA, "-i", fq, "-o", str(tmp_path / "ref"), "-t_db", prefix, "-t", "1"] + mode + opts, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _run(pre + ["-i", fq, "-t_db", prefix, "-o", str(tmp_path / "got")] + mode + opts, env=env)
        r, g = str(tmp_path / "ref"), str(tmp_path / "got")
        assert open(r + ".res").read() == open(g + ".res").read(), (mode, pre)
        assert open(r + ".fsa").read() == open(g + ".fsa").read(), (mode, pre)
        assert open(r + ".res").read().count("\n") > 8
