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
