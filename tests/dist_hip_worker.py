"""Worker of tests/test_dist_gpu.py (started by torch.distributed.run, one process per rank; gloo between the ranks, every rank
computing on the HIP path -- on cuda:LOCAL_RANK, or all on cuda:0 with KMA_SHARE_GPU=1 for a one-GPU box).

Read-sharded run of stages 2, 3a and 3b as DESIGN.md section 5 lays it out: each rank maps its contiguous read range, the two
ConClave score vectors are summed over the ranks, ConClave runs per shard on the global vectors, and its per-template outputs
are summed again. Rank 0 stores the result for the test to compare with a single-process run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("KMA_SHARE_GPU") == "1" else int(os.environ["LOCAL_RANK"])
    dist.init_process_group("gloo")
    from kma_amd import binding, formats, synth
    from kma_amd.dist import allreduce_scores, shard_bounds
    names, seqs = synth.make_gene_db(n_families=40, variants=5, seed=77)
    prefix = os.path.join(out_dir, "db")
    if rank == 0:
        formats.write_index(prefix, names, seqs)
    dist.barrier()
    reads, *_ = synth.make_reads(seqs, 20000, read_len=150, sub_rate=0.01, random_frac=0.02, seed=78)
    lo, hi = shard_bounds(len(reads), rank, world)
    batch = formats.pack_fixed(reads[lo:hi])
    db = binding.KmaHipDB(prefix, device=local)
    (rc_flag, flag, T_off, T), h = db.map_se(batch)
    aln = torch.from_numpy(h["alignment_scores"].astype(np.int64))
    uniq = torch.from_numpy(h["uniq_alignment_scores"].astype(np.int64))
    allreduce_scores(aln, uniq)
    h["alignment_scores"] = aln.numpy().astype(np.uint64)
    h["uniq_alignment_scores"] = uniq.numpy().astype(np.uint64)
    cc = db.conclave_se(batch.length, T_off, h)
    w = torch.from_numpy(cc["w_scores"].astype(np.int64))
    dp = torch.from_numpy(cc["depth"].astype(np.int64))
    allreduce_scores(w, dp)
    fc = torch.from_numpy(cc["fragment_counts"].astype(np.int64))
    rcnt = torch.from_numpy(cc["read_counts"].astype(np.int64))
    allreduce_scores(fc, rcnt)
    tm = [None] * world
    dist.all_gather_object(tm, cc["tmpl"])
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), aln=aln.numpy(), uniq=uniq.numpy(), w=w.numpy(), depth=dp.numpy(), frags=fc.numpy(),
                 reads=rcnt.numpy(), tmpl=np.concatenate(tm))
    db.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
