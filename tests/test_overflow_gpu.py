"""Capacity contract of the device-resident calls (kmahip.h, kmahip_ws_status): kmahip_scan_se_dev followed directly by
kmahip_align_se_dev on one stream with a T_cap that is too small must report KMAHIP_EOVERFLOW and touch nothing beyond the
capacities the caller gave (guard words behind every output array stay intact)."""
import numpy as np
import pytest

from kma_amd import formats, synth

pytestmark = pytest.mark.gpu

GUARD = 0x5A5A5A5A


def test_scan_dev_then_align_dev_with_small_T_cap_reports_overflow_and_stays_in_bounds(tmp_path):
    import torch
    from kma_amd import binding
    names, seqs = synth.make_gene_db(n_families=30, variants=5, seed=3)
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    reads, *_ = synth.make_reads(seqs, 4096, read_len=150, sub_rate=0.01, seed=5)
    b = formats.pack_fixed(reads)
    n = b.n
    dev = torch.device("cuda", 0)
    db = binding.KmaHipDB(prefix)
    try:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        seq = t(np.concatenate([b.seq, np.zeros(2, np.uint64)]).view(np.int64))
        seq_off, length, N_off = t(b.seq_off), t(b.length), t(b.N_off)
        N = t(b.N if len(b.N) else np.zeros(1, np.int32))
        cap = n // 2                      # every read has at least one candidate: far too small
        G = 64

        def guarded(m):
            x = torch.full((m + G,), GUARD, dtype=torch.int32, device=dev)
            return x
        rc_flag, flag, n_hits, best, oflag = (guarded(n) for _ in range(5))
        T, h_t, h_sc, h_s, h_e = (guarded(cap) for _ in range(5))
        T_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        aln = torch.zeros(int(db.info.DB_size), dtype=torch.int64, device=dev)
        uniq = torch.zeros_like(aln)
        db.scan_se_dev(seq, seq_off, length, N, N_off, rc_flag[:n], flag[:n], T_off, T[:cap])
        db.align_se_dev(seq, seq_off, length, N, N_off, 150, rc_flag[:n], flag[:n], T_off, T[:cap], n_hits[:n], best[:n], oflag[:n],
                        h_t[:cap], h_sc[:cap], h_s[:cap], h_e[:cap], aln, uniq)
        with pytest.raises(binding.KmaHipError, match="-6"):
            db.status()
        torch.cuda.synchronize()
        assert int(T_off[n].item()) > cap                      # the needed capacity is reported
        for x, m in ((rc_flag, n), (flag, n), (n_hits, n), (best, n), (oflag, n), (T, cap), (h_t, cap), (h_sc, cap), (h_s, cap), (h_e, cap)):
            assert bool((x[m:] == GUARD).all().item()), "a kernel wrote behind an output array"
        assert int(n_hits[:n].abs().sum().item()) == 0 and int(aln.sum().item()) == 0     # stage 3a reported no hits
        # the retry with the reported capacity works and the status is clean again
        cap2 = int(T_off[n].item())
        T2, h2 = torch.zeros(cap2, dtype=torch.int32, device=dev), [torch.zeros(cap2, dtype=torch.int32, device=dev) for _ in range(4)]
        db.scan_se_dev(seq, seq_off, length, N, N_off, rc_flag[:n], flag[:n], T_off, T2)
        db.align_se_dev(seq, seq_off, length, N, N_off, 150, rc_flag[:n], flag[:n], T_off, T2, n_hits[:n], best[:n], oflag[:n], *h2, aln, uniq)
        db.status()
        assert int((n_hits[:n] > 0).sum().item()) > n * 0.9
    finally:
        db.close()


def test_reads_full_of_repeats_grow_the_seed_capacity_and_equal_the_reference(tmp_path):
    """a gene that is a tandem repeat of a 24-base unit: a read out of it meets every one of its k-mers dozens of times in the
    template, i.e. hundreds of MEMs against one template -- more than the 64 slots per (read, template) the scratch starts with.
    The run raises the capacity and goes on (it used to end with KMAHIP_EOVERFLOW); files as the reference's."""
    import gzip
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(12)
    names, seqs = synth.make_gene_db(n_families=6, variants=3, seed=31)
    unit = rng.integers(0, 4, 24, dtype=np.uint8)
    rep = np.tile(unit, 50)
    rep[rng.integers(0, len(rep), 12)] = rng.integers(0, 4, 12, dtype=np.uint8)          # a few units differ
    seqs = list(seqs) + [np.concatenate([rng.integers(0, 4, 200, dtype=np.uint8), rep, rng.integers(0, 4, 200, dtype=np.uint8)])]
    names = list(names) + ["tandem"]
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads, *_ = synth.make_reads(seqs[:-1], 600, read_len=150, sub_rate=0.01, seed=5)
    reads = [r for r in reads]
    t = seqs[-1]
    for i in range(120):
        a = int(rng.integers(150, len(t) - 400))
        r = t[a:a + int(rng.integers(120, 260))].copy()
        r[rng.random(len(r)) < 0.01] = 0
        reads.append(r)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, reads, prefix="q")
    for mode in (["-1t1"], []):
        ref, got = str(tmp_path / ("ref" + "".join(mode))), str(tmp_path / ("got" + "".join(mode)))
        subprocess.run([KMA, "-i", fq, "-o", ref, "-t_db", prefix, "-t", "1"] + mode, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        g = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", got] + mode, stderr=subprocess.PIPE,
                           env=dict(os.environ, KMAHIP_DEBUG_TIMING="1"))
        assert g.returncode == 0, g.stderr.decode()[-500:]
        assert b"capacity per read and template raised" in g.stderr, mode          # (the case is what it claims to be)
        assert open(got + ".res", "rb").read() == open(ref + ".res", "rb").read(), mode
        assert b"tandem" in open(got + ".res", "rb").read()
        assert open(got + ".fsa", "rb").read() == open(ref + ".fsa", "rb").read(), mode
        assert gzip.open(got + ".frag.gz").read() == gzip.open(ref + ".frag.gz").read(), mode


def test_long_reads_full_of_repeats_grow_the_pipelines_seed_capacity_and_equal_the_reference(tmp_path):
    """the same for reads over 1 kb, whose stage 3a and traceback go through the long-read pipeline (longtrace.hip): a wavefront there
    starts with 1 024 MEM slots; reads of 1.2-2.5 kb out of a tandem repeat of 300 units carry thousands of MEMs. The pipeline raises
    its capacity and seeds the pass again (it used to end the run with KMAHIP_EOVERFLOW)."""
    import gzip
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    KMA = os.path.join(ROOT, "oracle", "_ref", "kma")
    if not os.path.exists(KMA):
        pytest.skip("oracle/_ref/kma not built")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(13)
    names, seqs = synth.make_gene_db(n_families=6, variants=3, len_lo=1500, len_hi=3000, seed=32)
    unit = rng.integers(0, 4, 24, dtype=np.uint8)
    rep = np.tile(unit, 300)
    rep[rng.integers(0, len(rep), 70)] = rng.integers(0, 4, 70, dtype=np.uint8)          # some units differ
    seqs = list(seqs) + [np.concatenate([rng.integers(0, 4, 300, dtype=np.uint8), rep, rng.integers(0, 4, 300, dtype=np.uint8)])]
    names = list(names) + ["tandem"]
    prefix = str(tmp_path / "db")
    synth.write_fasta(prefix + ".fsa", names, seqs)
    subprocess.run([KMA, "index", "-i", prefix + ".fsa", "-o", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = []
    for i in range(120):
        g = seqs[int(rng.integers(0, len(seqs) - 1))]
        reads.append(synth.make_long_reads(g, 1, read_len=int(rng.integers(1100, len(g))), seed=200 + i)[0])
    t = seqs[-1]
    for i in range(40):
        a = int(rng.integers(100, len(t) - 2800))
        r = t[a:a + int(rng.integers(1200, 2500))].copy()
        x = rng.random(len(r)) < 0.01
        r[x] = (r[x] + 1) & 3
        reads.append(r)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, reads, prefix="q")
    for mode in (["-1t1"], []):
        ref, got = str(tmp_path / ("ref" + "".join(mode))), str(tmp_path / ("got" + "".join(mode)))
        subprocess.run([KMA, "-i", fq, "-o", ref, "-t_db", prefix, "-t", "1"] + mode, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        g = subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", got] + mode, stderr=subprocess.PIPE,
                           env=dict(os.environ, KMAHIP_DEBUG_TIMING="1"))
        assert g.returncode == 0, g.stderr.decode()[-500:]
        assert b"longtrace: seed (MEM) capacity per read raised" in g.stderr, mode          # (the case is what it claims to be)
        assert open(got + ".res", "rb").read() == open(ref + ".res", "rb").read(), mode
        assert b"tandem" in open(got + ".res", "rb").read()
        assert open(got + ".fsa", "rb").read() == open(ref + ".fsa", "rb").read(), mode
        assert gzip.open(got + ".frag.gz").read() == gzip.open(ref + ".frag.gz").read(), mode
