"""N > 1 path on CPU: two gloo ranks map disjoint read shards (oracle as the compute stand-in),
all-reduce the ConClave vectors, run ConClave per shard on the summed vectors, reduce its per-template outputs, and must
reproduce the single-process result exactly."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util
from kma_amd import formats
from kma_amd.dist import allreduce_scores, shard_bounds


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, tmpdir, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    g = golden_util.load_se(os.path.join(tmpdir, f"r{rank}"))
    lo, hi = shard_bounds(len(g["reads"]), rank, world)
    batch = formats.pack_ragged(g["reads"][lo:hi])
    db = oracle.OracleDB(g["prefix"])
    sc = db.scan_se(batch)
    res = db.align_se(batch, *sc)
    aln = torch.from_numpy(res["alignment_scores"].astype(np.int64))
    uniq = torch.from_numpy(res["uniq_alignment_scores"].astype(np.int64))
    mapped = torch.tensor([int((res["n_hits"] > 0).sum())])
    allreduce_scores(aln, uniq)
    dist.all_reduce(mapped)
    # stage 3b on the shard with the GLOBAL vectors, then the second reduction: w_scores and depth
    tlen = formats.read_lengths(g["prefix"])
    cc = oracle.conclave(res["n_hits"], res["best_score"], batch.length, np.zeros(batch.n, np.int32), sc[2][:-1], res["tmpl"],
                         res["start"], res["end"], aln.numpy().astype(np.uint64), uniq.numpy().astype(np.uint64), tlen)
    w = torch.from_numpy(cc["w_scores"].astype(np.int64))
    dp = torch.from_numpy(cc["depth"].astype(np.int64))
    allreduce_scores(w, dp)
    if rank == 0:
        np.save(out + ".w", np.stack([w.numpy(), dp.numpy()]))
        np.save(out, np.stack([aln.numpy(), uniq.numpy()]))
        open(out + ".mapped", "w").write(str(int(mapped)))
    dist.destroy_process_group()


def test_two_rank_read_shards_allreduce_matches_single_process(tmp_path):
    import oracle
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for r in range(2):
        os.makedirs(tmp_path / f"r{r}")
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    got = np.load(out)
    os.makedirs(tmp_path / "single")
    g = golden_util.load_se(tmp_path / "single")
    db = oracle.OracleDB(g["prefix"])
    sc = db.scan_se(g["batch"])
    res = db.align_se(g["batch"], *sc)
    assert np.array_equal(got[0], res["alignment_scores"].astype(np.int64))
    assert np.array_equal(got[1], res["uniq_alignment_scores"].astype(np.int64))
    assert int(open(out + ".mapped").read()) == int((res["n_hits"] > 0).sum()) == 966
    # ConClave per shard over the summed vectors + summed outputs == ConClave over the whole stream
    tlen = formats.read_lengths(g["prefix"])
    b = g["batch"]
    cc = oracle.conclave(res["n_hits"], res["best_score"], b.length, np.zeros(b.n, np.int32), sc[2][:-1], res["tmpl"], res["start"],
                         res["end"], res["alignment_scores"], res["uniq_alignment_scores"], tlen)
    w = np.load(out + ".w.npy")
    assert np.array_equal(w[0], cc["w_scores"].astype(np.int64)) and np.array_equal(w[1], cc["depth"].astype(np.int64))


def test_template_owners_are_contiguous_and_balanced():
    from kma_amd.dist import template_owners
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        c = rng.integers(0, 50, 200)
        c[rng.random(200) < 0.5] = 0
        o = template_owners(c, world)
        assert np.all(np.diff(o) >= 0) and o.min() == 0 and o.max() <= world - 1
        per = np.bincount(o, weights=c, minlength=world)
        assert per.max() <= c.sum() / world + c.max()
    assert np.array_equal(template_owners(np.zeros(5, np.int64), 4), np.zeros(5, np.int64))


class _Placed(torch.Tensor):
    """marks a tensor that kma_amd.dist._place made"""


def _guard_collectives():
    """Every tensor a collective of kma_amd.dist receives must have been placed for the group's backend by dist._place: under
    nccl (= RCCL) a CPU tensor raises "No backend type associated with device type cpu", which no gloo run would ever show.
    _place is made to tag what it returns and the collectives are wrapped to reject untagged tensors."""
    import kma_amd.dist as kd
    place = kd._place
    kd._place = lambda t, group=None: place(t, group).as_subclass(_Placed)
    seen = []

    def wrap(name):
        real = getattr(dist, name)

        def checked(*a, **k):
            flat = [x for arg in list(a) + list(k.values()) for x in (arg if isinstance(arg, (list, tuple)) else [arg])]
            ts = [x for x in flat if isinstance(x, torch.Tensor)]
            assert ts and all(isinstance(x, _Placed) for x in ts), f"dist.{name} received a tensor that did not come from dist._place"
            seen.append(name)
            return real(*[x.as_subclass(torch.Tensor) if isinstance(x, _Placed) else
                          [y.as_subclass(torch.Tensor) for y in x] if isinstance(x, list) and x and isinstance(x[0], torch.Tensor) else x for x in a], **k)
        setattr(dist, name, checked)
    for name in ("all_reduce", "all_gather", "all_to_all_single", "all_gather_into_tensor", "broadcast", "reduce_scatter_tensor"):
        wrap(name)
    return seen


def _gather_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kma_amd.dist import all_reduce_sum, gather_filed_reads, shard_bounds as sb
    seen = _guard_collectives()
    # the guard itself: an unplaced tensor is refused, the helpers pass
    try:
        dist.all_reduce(torch.zeros(2, dtype=torch.int64))
        raise SystemExit("the guard let an unplaced tensor through")
    except AssertionError:
        pass
    assert int(all_reduce_sum(torch.tensor([rank + 1]))[0]) == world * (world + 1) // 2
    a, u = torch.tensor([1, 2 + rank]), torch.tensor([3, 4])
    allreduce_scores(a, u)
    assert a.tolist() == [world, sum(2 + r for r in range(world))] and u.tolist() == [3 * world, 4 * world]
    reads, tmpl, stats, runs, names, nh, rc = _gather_case()
    lo, hi = sb(len(reads), rank, world)
    batch = formats.pack_ragged(reads[lo:hi])
    n_ops = np.array([len(r) for r in runs[lo:hi]], np.int32)
    ops_off = np.zeros(hi - lo, np.int64)
    ops_off[1:] = np.cumsum(n_ops[:-1])
    ops = np.concatenate(runs[lo:hi]).astype(np.uint32)
    owner = np.array([0, 0, 0, 1, 1, 0, 1, 1], np.int64)              # (any map works for the exchange; contiguity is the caller's business)
    got = gather_filed_reads(batch, rc[lo:hi], tmpl[lo:hi], nh[lo:hi], (stats[lo:hi], ops_off, n_ops, ops), owner, names[lo:hi])
    b, rc2, tm2, nh2, tr2, rank2, names2 = got
    np.savez(f"{out}.{rank}.npz", seq=b.seq, seq_off=b.seq_off, length=b.length, N=b.N, N_off=b.N_off, rc=rc2, tmpl=tm2, nh=nh2, stats=tr2[0],
             ops_off=tr2[1], n_ops=tr2[2], ops=tr2[3], rank=rank2, names=np.array(names2, dtype=object))
    assert {"all_reduce", "all_gather", "all_to_all_single"} <= set(seen)
    dist.destroy_process_group()


def _gather_case():
    rng = np.random.default_rng(11)
    n = 301
    reads = [rng.integers(0, 5, int(rng.integers(20, 200)), dtype=np.uint8) for _ in range(n)]
    tmpl = rng.integers(-7, 8, n).astype(np.int32)
    stats = rng.integers(1, 100, (n, 10)).astype(np.int32)
    stats[rng.random(n) < 0.2, 3] = 0                                   # dropped by the stage-3c filter: filed, but no pile-up and no row
    runs = [rng.integers(1, 1 << 20, int(rng.integers(1, 6))).astype(np.uint32) for _ in range(n)]
    names = [b"read%d/x" % i for i in range(n)]
    return reads, tmpl, stats, runs, names, rng.integers(1, 4, n).astype(np.int32), rng.integers(0, 2, n).astype(np.int32)


def test_filed_reads_reach_their_template_owner_in_stream_order(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "got")
    mp.spawn(_gather_worker, args=(2, port, out), nprocs=2, join=True)
    reads, tmpl, stats, runs, names, nh, rc = _gather_case()
    owner = np.array([0, 0, 0, 1, 1, 0, 1, 1], np.int64)
    filed = tmpl != 0
    frag_rank = np.cumsum(filed) - filed
    for r in range(2):
        g = np.load(f"{out}.{r}.npz", allow_pickle=True)
        want = [i for i in range(len(reads)) if filed[i] and stats[i, 3] != 0 and owner[abs(tmpl[i])] == r]
        assert len(g["length"]) == len(want) > 20
        assert np.array_equal(g["rank"], frag_rank[want]) and np.array_equal(g["tmpl"], tmpl[want])
        assert np.array_equal(g["rc"], rc[want]) and np.array_equal(g["nh"], nh[want]) and np.array_equal(g["stats"], stats[want])
        assert list(g["names"]) == [names[i] for i in want]
        ref = formats.pack_ragged([reads[i] for i in want])
        for k in ("seq", "seq_off", "length", "N", "N_off"):
            assert np.array_equal(g[k], getattr(ref, k)), k
        for j, i in enumerate(want):
            assert np.array_equal(g["ops"][g["ops_off"][j]:g["ops_off"][j] + g["n_ops"][j]], runs[i])
