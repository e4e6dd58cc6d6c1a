"""Synthetic read batches generated directly in HBM with torch (benchmark plumbing).

Same distribution as synth.make_reads (SURVEY.md §8d "Reads-150"): gene uniform,
start uniform, 0.5 % substitutions, 50 % reverse-complemented, optional random
reads; packed into the kmahip_reads CSR layout (5 words + 1 pad per 150 bp read).
"""
from __future__ import annotations

import os

import numpy as np
import torch


def make_packed_reads(seqs, n_reads, read_len=150, sub_rate=0.005, rc_frac=0.5, random_frac=0.0,
                      seed=1, device="cuda", chunk=1 << 20, keep_codes=0, junk_frac=0.0):
    """-> dict(seq i64[n*(W+1)], seq_off i64[n+1], length i32[n], N i32[1], N_off i64[n+1], codes u8[keep_codes, L])"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = np.array([len(s) for s in seqs], np.int64)
    ok = np.nonzero(lens >= read_len)[0]
    cat = torch.from_numpy(np.concatenate(seqs)).to(device)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])[:-1]).to(device)
    okt = torch.from_numpy(ok).to(device)
    lens_t = torch.from_numpy(lens).to(device)
    W = (read_len + 31) // 32
    out = torch.zeros((n_reads, W + 1), dtype=torch.int64, device=device)
    shifts = (62 - 2 * torch.arange(32, device=device, dtype=torch.int64))
    ar = torch.arange(read_len, device=device)
    codes_keep = []
    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        gi = okt[torch.randint(0, len(ok), (m,), generator=g, device=device)]
        st = (torch.rand(m, generator=g, device=device, dtype=torch.float64) * (lens_t[gi] - read_len + 1).double()).long()
        if os.environ.get("KMAHIP_SYNTH_SORTED"):          # locality experiment only: reads in template order
            order = torch.argsort(offs[gi] + st)
            gi, st = gi[order], st[order]
        idx = (offs[gi] + st)[:, None] + ar[None, :]
        r = cat[idx]
        if sub_rate > 0:
            mut = torch.rand(r.shape, generator=g, device=device) < sub_rate
            add = torch.randint(1, 4, r.shape, generator=g, device=device, dtype=torch.uint8)
            r = torch.where(mut, (r + add) & 3, r)
        rc = torch.rand(m, generator=g, device=device) < rc_frac
        r = torch.where(rc[:, None], 3 - r.flip(1), r)
        if junk_frac > 0:
            # reads that match their gene only in part: the last 60 ... 120 bases replaced by foreign sequence (adapters,
            # chimeras) -- the unaligned-end DP problems of stage 3a
            jk = torch.rand(m, generator=g, device=device) < junk_frac
            cut = read_len - torch.randint(60, 121, (m,), generator=g, device=device)
            rr = torch.randint(0, 4, r.shape, generator=g, device=device, dtype=torch.uint8)
            r = torch.where(jk[:, None] & (ar[None, :] >= cut[:, None]), rr, r)
        if random_frac > 0:
            rnd = torch.rand(m, generator=g, device=device) < random_frac
            rr = torch.randint(0, 4, r.shape, generator=g, device=device, dtype=torch.uint8)
            r = torch.where(rnd[:, None], rr, r)
        if len(codes_keep) * chunk < keep_codes:
            codes_keep.append(r[: max(0, keep_codes - c0)].cpu())
        pad = W * 32 - read_len
        r64 = r.long()
        if pad:
            r64 = torch.cat([r64, torch.zeros((m, pad), dtype=torch.int64, device=device)], dim=1)
        out[c0:c0 + m, :W] = (r64.view(m, W, 32) << shifts[None, None, :]).sum(dim=2)
        del idx, r, r64
    n = n_reads
    return dict(
        seq=out.view(-1),
        seq_off=torch.arange(n + 1, device=device, dtype=torch.int64) * (W + 1),
        length=torch.full((n,), read_len, dtype=torch.int32, device=device),
        N=torch.zeros(1, dtype=torch.int32, device=device),
        N_off=torch.zeros(n + 1, dtype=torch.int64, device=device),
        codes=(torch.cat(codes_keep).numpy() if codes_keep else None),
    )


def iter_pair_codes(seqs, n_pairs, read_len=150, ins_lo=250, ins_hi=450, sub_rate=0.005, seed=1, device="cuda", chunk=1 << 20):
    """Pairs of the shape of synth.make_pairs (mate 1 = the fragment's first read_len bases, mate 2 = the reverse complement of its
    last read_len, insert U[ins_lo, ins_hi] clipped to the gene, half of the fragments sequenced from the other strand, 0.5 %
    substitutions), generated in HBM: yields (first pair, u8[2 m, read_len]) with the mates INTERLEAVED (rows 2i, 2i + 1 = pair i)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = np.array([len(s) for s in seqs], np.int64)
    ok = np.nonzero(lens >= read_len)[0]
    cat = torch.from_numpy(np.concatenate(seqs)).to(device)
    offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)])[:-1]).to(device)
    okt = torch.from_numpy(ok).to(device)
    lens_t = torch.from_numpy(lens).to(device)
    ar = torch.arange(read_len, device=device)
    for c0 in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - c0)
        gi = okt[torch.randint(0, len(ok), (m,), generator=g, device=device)]
        ins = torch.minimum(torch.randint(ins_lo, ins_hi + 1, (m,), generator=g, device=device), lens_t[gi])
        st = (torch.rand(m, generator=g, device=device, dtype=torch.float64) * (lens_t[gi] - ins + 1).double()).long()
        a = cat[(offs[gi] + st)[:, None] + ar[None, :]]
        b = 3 - cat[(offs[gi] + st + ins - read_len)[:, None] + ar[None, :]].flip(1)
        sw = torch.rand(m, generator=g, device=device) < 0.5
        both = torch.stack([torch.where(sw[:, None], b, a), torch.where(sw[:, None], a, b)], dim=1).view(2 * m, read_len)
        if sub_rate > 0:
            mut = torch.rand(both.shape, generator=g, device=device) < sub_rate
            add = torch.randint(1, 4, both.shape, generator=g, device=device, dtype=torch.uint8)
            both = torch.where(mut, (both + add) & 3, both)
        yield c0, both


def make_packed_pairs(seqs, n_pairs, read_len=150, seed=1, device="cuda", chunk=1 << 20, keep_codes=0, **kw):
    """iter_pair_codes packed into the kmahip_reads CSR layout with the mates interleaved (read 2i, 2i + 1 = pair i: what
    kmahip_scan_pe takes). -> dict like make_packed_reads; codes = (m1 u8[keep, L], m2 u8[keep, L]) of the first keep_codes pairs."""
    W = (read_len + 31) // 32
    n = 2 * n_pairs
    out = torch.zeros((n, W + 1), dtype=torch.int64, device=device)
    shifts = (62 - 2 * torch.arange(32, device=device, dtype=torch.int64))
    keep1, keep2 = [], []
    for c0, both in iter_pair_codes(seqs, n_pairs, read_len=read_len, seed=seed, device=device, chunk=chunk, **kw):
        m = both.shape[0] // 2
        if c0 < keep_codes:
            k = min(m, keep_codes - c0)
            keep1.append(both[0:2 * k:2].cpu()); keep2.append(both[1:2 * k:2].cpu())
        pad = W * 32 - read_len
        r64 = both.long()
        if pad:
            r64 = torch.cat([r64, torch.zeros((2 * m, pad), dtype=torch.int64, device=device)], dim=1)
        out[2 * c0:2 * (c0 + m), :W] = (r64.view(2 * m, W, 32) << shifts[None, None, :]).sum(dim=2)
        del both, r64
    return dict(
        seq=out.view(-1),
        seq_off=torch.arange(n + 1, device=device, dtype=torch.int64) * (W + 1),
        length=torch.full((n,), read_len, dtype=torch.int32, device=device),
        N=torch.zeros(1, dtype=torch.int32, device=device),
        N_off=torch.zeros(n + 1, dtype=torch.int64, device=device),
        codes=((torch.cat(keep1).numpy(), torch.cat(keep2).numpy()) if keep1 else None),
    )


def make_long_reads_packed(genome, n_reads, read_len=10000, sub=0.04, dele=0.03, ins=0.03, rc_frac=0.5, seed=8, device="cuda",
                           chunk=2048, keep_codes=0):
    """ONT-like reads of one genome (SURVEY.md §8d "ONT-10k": windows of read_len bases, substitutions / deletions / insertions,
    half of them reverse-complemented), generated and 2-bit packed on the device, returned as HOST arrays in the kmahip_reads CSR
    layout (each read followed by one pad word): dict(seq u64, seq_off i64[n+1], length i32[n], N i32[1], N_off i64[n+1],
    codes = the first keep_codes reads as a list of uint8 arrays -- what the reference is given as FASTQ for the parity subset).
    Same distribution as synth.make_long_reads, not the same random stream."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    G = len(genome)
    gen = torch.from_numpy(np.ascontiguousarray(genome)).to(device)
    ar = torch.arange(read_len, device=device)
    seqs, lens, codes = [], [], []
    for c0 in range(0, n_reads, chunk):
        m = min(chunk, n_reads - c0)
        st = torch.randint(0, G - read_len + 1, (m,), generator=g, device=device)
        w = gen[st[:, None] + ar[None, :]]
        mut = torch.rand((m, read_len), generator=g, device=device) < sub
        w = torch.where(mut, (w + torch.randint(1, 4, (m, read_len), generator=g, device=device, dtype=torch.uint8)) & 3, w)
        keep = torch.rand((m, read_len), generator=g, device=device) >= dele
        insb = (torch.rand((m, read_len), generator=g, device=device) < ins) & keep     # a random base in front of a kept base
        cnt = keep.to(torch.int32) + insb.to(torch.int32)
        end = torch.cumsum(cnt, 1)
        L = end[:, -1].to(torch.int64)                                                  # read lengths
        off = (end - cnt).to(torch.int64)
        rc = torch.rand(m, generator=g, device=device) < rc_frac
        nw = ((L + 31) >> 5) + 1
        woff = torch.cumsum(nw, 0) - nw
        flat = torch.zeros(int(nw.sum()), dtype=torch.int64, device=device)
        pad = torch.zeros((m, int(L.max()) if keep_codes > c0 else 1), dtype=torch.uint8, device=device)

        def put(sel, pos, val):
            # base `val` of read row at read position pos (before the strand turn)
            rows = sel.nonzero(as_tuple=True)[0]
            q = pos[sel]
            v = val[sel].to(torch.int64)
            turn = rc[rows]
            q = torch.where(turn, L[rows] - 1 - q, q)
            v = torch.where(turn, 3 - v, v)
            flat.index_add_(0, woff[rows] + (q >> 5), v << (62 - 2 * (q & 31)))
            if keep_codes > c0:
                pad[rows, q] = v.to(torch.uint8)
        put(insb, off, torch.randint(0, 4, (m, read_len), generator=g, device=device, dtype=torch.uint8))
        put(keep, off + insb.to(torch.int64), w)
        seqs.append(flat.cpu().numpy().view(np.uint64))
        lens.append(L.cpu().numpy())
        if keep_codes > c0:
            hp, hl = pad.cpu().numpy(), lens[-1]
            for i in range(min(m, keep_codes - c0)):
                codes.append(hp[i, :hl[i]].copy())
        del w, mut, keep, insb, cnt, end, off, flat, pad
    length = np.concatenate(lens).astype(np.int32)
    nw = ((length.astype(np.int64) + 31) >> 5) + 1
    seq_off = np.zeros(n_reads + 1, np.int64)
    seq_off[1:] = np.cumsum(nw)
    return dict(seq=np.concatenate(seqs), seq_off=seq_off, length=length, N=np.zeros(1, np.int32), N_off=np.zeros(n_reads + 1, np.int64), codes=codes)
