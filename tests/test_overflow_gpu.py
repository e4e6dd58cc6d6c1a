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
