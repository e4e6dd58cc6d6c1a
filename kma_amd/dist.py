"""Read sharding across ranks and the path's single exchange step.

Stages 2 and 3a are independent per read, so each rank (one process per GPU)
maps a contiguous range of reads against its own copy of the database in HBM.
The only data that crosses ranks is the SUM of the two per-template ConClave
vectors alignment_scores / uniq_alignment_scores (u64[DB_size],
updatescores.c:228,276) which runConClave consumes (runkma.c:563-594).
backend "nccl" is RCCL on ROCm; tests run the same code over gloo on CPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_reads: int, rank: int, world: int):
    """Contiguous, balanced read range [lo, hi) of `rank`."""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _device(group=None):
    """Where a collective's tensors have to live for the backend of `group`: RCCL moves device memory only (a CPU tensor
    under nccl raises "No backend type associated with device type cpu"), gloo host memory."""
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _place(t: torch.Tensor, group=None) -> torch.Tensor:
    """`t` where the backend of `group` can reach it. EVERY tensor handed to a collective in this module -- inputs and
    receive buffers -- is made by this function (tests/test_dist_gloo.py wraps the collectives and rejects any other)."""
    dev = _device(group)
    return t if t.device == dev else t.to(dev)


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """SUM of `t` over the ranks of `group`, returned on the device `t` came from."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    x = _place(t, group)
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
    return x.to(t.device)


def all_gather_ints(value: int, group=None):
    """[value of rank 0, value of rank 1, ...] (a tensor collective: no pickling, placed like the others)."""
    world = dist.get_world_size(group)
    mine = _place(torch.tensor([int(value)], dtype=torch.int64), group)
    out = [_place(torch.empty(1, dtype=torch.int64), group) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(x.cpu()) for x in out]


def allreduce_scores(alignment_scores: torch.Tensor, uniq_alignment_scores: torch.Tensor, group=None):
    """In-place SUM over ranks of the two score vectors (int64 view of the u64 sums: exact, order-free)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    both = all_reduce_sum(torch.stack([alignment_scores, uniq_alignment_scores]), group)
    alignment_scores.copy_(both[0])
    uniq_alignment_scores.copy_(both[1])


# ---- the whole single-end pipeline over read shards ---------------------------------------
# Stages 2, 3a, 3b and the traceback of 3c work on a rank's own reads; three things cross ranks:
#   1. SUM of alignment_scores / uniq_alignment_scores before ConClave (updatescores.c:228,276 -> conclave.c:89-113);
#   2. SUM of ConClave's per-template outputs (w_scores, conclave.c:147; fragment / read counts, depth) before the `.res`
#      statistics (runkma.c:608-613, 770-783) -- every rank then computes the same rows and the same set of templates to assemble;
#   3. the traced reads travel to the rank that OWNS their template (contiguous template ranges, balanced by filed fragments),
#      arriving in the order of the global stream with their positions among the filed fragments of the whole stream, which is
#      what the reference's assembly order is made of (conclave.c:164-166, 194; assembly.c:1377-1424).
# The owners pile up and call the consensus of their templates; per-template results are disjoint between owners.

import numpy as np

from .formats import ReadBatch


def template_owners(fragment_counts: np.ndarray, world: int) -> np.ndarray:
    """owner[t]: contiguous template ranges in template order, cut where the filed fragments before a template reach the next
    1 / world of the total (templates without fragments go with their neighbours)."""
    c = np.asarray(fragment_counts, np.int64)
    before = np.cumsum(c) - c
    total = int(c.sum())
    if total == 0:
        return np.zeros(len(c), np.int64)
    return np.minimum(before * world // total, world - 1).astype(np.int64)


def exchange(parts, dtype, group=None):
    """parts[d] = 1-D numpy array for rank d -> list over source ranks of what they sent here (all_to_all_single with the
    sizes exchanged first)."""
    world = dist.get_world_size(group)
    send = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype).ravel() for p in parts]) if world else np.zeros(0, dtype), dtype)
    n_in = _place(torch.tensor([len(np.asarray(p).ravel()) for p in parts], dtype=torch.int64), group)
    n_out = _place(torch.empty(world, dtype=torch.int64), group)
    dist.all_to_all_single(n_out, n_in, group=group)
    n_out_l = [int(x) for x in n_out.cpu()]
    # (torch has no unsigned 64-bit collectives: the bytes travel as they are)
    tsend = _place(torch.from_numpy(send.view(np.uint8).copy()), group)
    item = np.dtype(dtype).itemsize
    trecv = _place(torch.empty(sum(n_out_l) * item, dtype=torch.uint8), group)
    dist.all_to_all_single(trecv, tsend, [x * item for x in n_out_l], [int(x) * item for x in n_in.cpu()], group=group)
    flat = trecv.cpu().numpy().view(dtype)
    out, at = [], 0
    for x in n_out_l:
        out.append(flat[at:at + x])
        at += x
    return out


def _csr_take(data, starts, lens):
    """The concatenation of data[starts[i] : starts[i] + lens[i]] over i, as ONE fancy-index gather (no per-read loop)."""
    lens = np.asarray(lens, np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, np.asarray(data).dtype)
    before = np.cumsum(lens) - lens
    return np.asarray(data)[np.repeat(np.asarray(starts, np.int64) - before, lens) + np.arange(total)]


def _names_blob(names):
    """list[bytes] -> (u8 blob, int64 offsets[n + 1]); a (blob, offsets) pair passes through."""
    if isinstance(names, tuple):
        return np.asarray(names[0], np.uint8), np.asarray(names[1], np.int64)
    off = np.zeros(len(names) + 1, np.int64)
    if len(names):
        off[1:] = np.cumsum(np.fromiter(map(len, names), np.int64, len(names)))
    return np.frombuffer(b"".join(names), np.uint8), off


def gather_filed_reads(batch: ReadBatch, rc, tmpl, n_hits, traces, owner, names=None, group=None):
    """Step 3 above. Every rank passes its shard (reads, strand flags, ConClave templates, tie counts, the tuple align_trace
    returned); returns what this rank owns: (ReadBatch, rc, tmpl, n_hits, traces, frag_rank, names) with the reads in the order
    of the global stream. names: list[bytes] or (u8 blob, offsets[n + 1]); it comes back as a list. All gathers are array
    operations -- the cost is a few passes over the kept reads' bytes, not interpreter time per read."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    stats, ops_off, n_ops, ops = traces
    n = batch.n
    tmpl = np.asarray(tmpl, np.int32)
    filed = tmpl != 0
    # position among the filed fragments of the whole stream: shards are contiguous in stream order
    counts = all_gather_ints(int(filed.sum()), group)
    base = int(sum(counts[:rank]))
    frag_rank = base + np.cumsum(filed) - filed
    kept = np.nonzero(filed & (np.asarray(stats)[:, 3] != 0))[0] if n else np.zeros(0, np.int64)
    dest = owner[np.abs(tmpl[kept])] if len(kept) else np.zeros(0, np.int64)
    # kept reads grouped by destination, stream order inside a group (a stable sort of a few small integers)
    order = kept[np.argsort(dest, kind="stable")]
    per_dest = np.bincount(dest, minlength=world)[:world] if len(kept) else np.zeros(world, np.int64)
    cut = np.zeros(world + 1, np.int64)
    cut[1:] = np.cumsum(per_dest)
    seq_off, N_off = np.asarray(batch.seq_off, np.int64), np.asarray(batch.N_off, np.int64)
    L = np.asarray(batch.length, np.int64)[order]
    nw = (L + 31) >> 5
    nN = N_off[order + 1] - N_off[order]
    nops = np.asarray(n_ops, np.int64)[order]
    if names is not None:
        nblob, noff = _names_blob(names)
        nlen = noff[order + 1] - noff[order]
    else:
        nlen = np.zeros(len(order), np.int64)
    tab = np.zeros((len(order), 18), np.int64)
    if len(order):
        tab[:, 0] = frag_rank[order]; tab[:, 1] = L; tab[:, 2] = np.asarray(rc)[order]; tab[:, 3] = tmpl[order]
        tab[:, 4] = np.asarray(n_hits)[order]; tab[:, 5] = nN; tab[:, 6] = nops; tab[:, 7] = nlen
        tab[:, 8:18] = np.asarray(stats)[order]
    words = _csr_take(batch.seq, seq_off[order], nw).astype(np.uint64, copy=False)
    npos = _csr_take(batch.N, N_off[order], nN).astype(np.int32, copy=False)
    runs = _csr_take(ops, np.asarray(ops_off, np.int64)[order], nops).astype(np.uint32, copy=False)
    nbytes = _csr_take(nblob, noff[order], nlen) if names is not None else np.zeros(0, np.uint8)

    def cuts(per_read):
        c = np.zeros(len(order) + 1, np.int64)
        c[1:] = np.cumsum(per_read)
        return [slice(int(c[cut[d]]), int(c[cut[d + 1]])) for d in range(world)]
    got_fixed = exchange([tab[cut[d]:cut[d + 1]].ravel() for d in range(world)], np.int64, group)
    got_words = exchange([words[s] for s in cuts(nw)], np.uint64, group)
    got_npos = exchange([npos[s] for s in cuts(nN)], np.int32, group)
    got_runs = exchange([runs[s] for s in cuts(nops)], np.uint32, group)
    got_names = exchange([nbytes[s] for s in cuts(nlen)], np.uint8, group)
    tab = np.concatenate(got_fixed).reshape(-1, 18) if got_fixed else np.zeros((0, 18), np.int64)
    m = len(tab)
    L = tab[:, 1].astype(np.int32)
    nw = ((L.astype(np.int64) + 31) >> 5)
    w_in = np.concatenate(got_words) if m else np.zeros(0, np.uint64)
    # every read followed by one pad word
    seq_off2 = np.zeros(m + 1, np.int64)
    seq_off2[1:] = np.cumsum(nw + 1)
    seq2 = np.zeros(int(seq_off2[-1]) if m else 1, np.uint64)
    src = np.zeros(m + 1, np.int64)
    src[1:] = np.cumsum(nw)
    if m:
        # scatter the words of each read to its padded place
        pos = np.repeat(seq_off2[:-1] - src[:-1], nw) + np.arange(int(src[-1]))
        seq2[pos] = w_in
    N_off2 = np.zeros(m + 1, np.int64)
    N_off2[1:] = np.cumsum(tab[:, 5])
    N2 = np.concatenate(got_npos).astype(np.int32) if m and int(N_off2[-1]) else np.zeros(0, np.int32)
    ops_off2 = np.zeros(m, np.int64)
    if m:
        ops_off2[1:] = np.cumsum(tab[:-1, 6])
    ops2 = np.concatenate(got_runs).astype(np.uint32) if m else np.zeros(0, np.uint32)
    names2 = None
    if names is not None:
        blob = np.concatenate(got_names).tobytes() if m else b""
        ends = np.cumsum(tab[:, 7]).tolist()
        names2 = [blob[a:b] for a, b in zip([0] + ends[:-1], ends)]
    b2 = ReadBatch(seq2, seq_off2, L, N2, N_off2)
    tr2 = (np.ascontiguousarray(tab[:, 8:18], dtype=np.int32), ops_off2, tab[:, 6].astype(np.int32), ops2)
    return b2, tab[:, 2].astype(np.int32), tab[:, 3].astype(np.int32), tab[:, 4].astype(np.int32), tr2, tab[:, 0].copy(), names2


def run_se_sharded(db, batch: ReadBatch, names=None, evalue=0.05, bcd=1, max_frag=0, frag_path=None, group=None):
    """The single-end `-1t1` run of kmahip_run_se with the reads sharded over the ranks of `group` (one process per GPU, `db` =
    this rank's KmaHipDB, `batch` = this rank's contiguous part of the stream). Returns, on every rank, dict(rows = the `.res`
    statistics of all templates (identical everywhere), owner [DB_size], cover / aln_len / depth / asm_len [DB_size] summed over
    the owners, consensus {template: str} of the templates this rank owns, frag_rows). frag_path: each rank writes the
    `.frag(.gz)` rows of its templates to frag_path % rank; concatenated in rank order they are the reference's file."""
    world = dist.get_world_size(group)
    (rc_flag, flag, T_off, T), h = db.map_se(batch)
    aln = torch.from_numpy(h["alignment_scores"].astype(np.int64))
    uniq = torch.from_numpy(h["uniq_alignment_scores"].astype(np.int64))
    allreduce_scores(aln, uniq, group)                                                     # exchange 1
    h["alignment_scores"] = aln.numpy().astype(np.uint64)
    h["uniq_alignment_scores"] = uniq.numpy().astype(np.uint64)
    cc = db.conclave_se(batch.length, T_off, h)
    per_t = torch.from_numpy(np.stack([cc[k].astype(np.int64) for k in ("w_scores", "depth", "fragment_counts", "read_counts")]))
    per_t = all_reduce_sum(per_t, group)                                                   # exchange 2
    w_scores = per_t[0].numpy().astype(np.uint64)
    rows = db.res_rows(w_scores, evalue=evalue)
    D = int(db.info.DB_size)
    ok = np.zeros(D + 8, np.uint8)
    for r in rows:
        ok[r.template_id] = 1 if r.significant else 0
    traces = db.align_trace(batch, h["rc"], cc["tmpl"], tmpl_ok=ok)
    owner = template_owners(per_t[2].numpy(), world)
    b2, rc2, tm2, nh2, tr2, rank2, names2 = gather_filed_reads(batch, h["rc"], cc["tmpl"], h["n_hits"], traces, owner, names, group)   # exchange 3
    asm = db.assemble(b2, rc2, tm2, tr2, max_frag=max_frag, bcd=bcd, evalue=evalue, consensus=True, frag_rank=rank2)
    frag_rows = 0
    if frag_path is not None and names2 is not None:
        frag_rows = db.frag_write2(frag_path % dist.get_rank(group), b2, rc2, tm2, nh2, tr2[0], names2, order=0, max_frag=max_frag, frag_rank=rank2)
    figs = torch.from_numpy(np.stack([asm[k].astype(np.int64) for k in ("cover", "aln_len", "depth", "asm_len")]))
    figs = all_reduce_sum(figs, group)                                                     # (owners are disjoint: a gather in template order)
    out = dict(rows=rows, owner=owner, consensus=asm.get("consensus", {}), frag_rows=frag_rows, tmpl=cc["tmpl"], depth_sum=per_t[1].numpy(),
               fragment_counts=per_t[2].numpy(), read_counts=per_t[3].numpy())
    for i, k in enumerate(("cover", "aln_len", "depth", "asm_len")):
        out[k] = figs[i].numpy()
    return out
