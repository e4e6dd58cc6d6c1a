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
