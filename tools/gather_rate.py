#!/usr/bin/env python3
"""Microbenchmark (GPU box): rate of random 32-byte row gathers from tables of several sizes (torch index_select) -- the
memory-system ceiling the probe kernels can be compared with.  usage: python3 tools/gather_rate.py"""
import torch

dev = torch.device("cuda", 0)
n = 64_000_000
for mb in (4, 16, 67, 256, 1024):
    rows = mb * 1024 * 1024 // 32
    table = torch.randint(0, 2**31 - 1, (rows, 8), dtype=torch.int32, device=dev)
    idx = torch.randint(0, rows, (n,), dtype=torch.int64, device=dev)
    out = torch.empty((n, 8), dtype=torch.int32, device=dev)
    for _ in range(2):
        torch.index_select(table, 0, idx, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        torch.index_select(table, 0, idx, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"table {mb:5d} MB: {n / ms / 1e6:7.1f} G gathers/s  ({ms:.2f} ms for {n} x 32 B rows; 32 B read + 32 B written + 8 B index each)")
    del table, idx, out
