"""Worker of tests/test_dist_gpu.py::test_sharded_whole_pipeline... (torch.distributed.run, gloo between the ranks, every rank on the
HIP path; KMA_SHARE_GPU=1 puts all ranks on cuda:0). The whole single-end run over read shards (kma_amd.dist.run_se_sharded):
rank 0 collects the `.res` lines, the consensus sequences and the `.frag` rows the ranks wrote for their templates."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def case():
    """a 200-gene database and a stream of reads: 150-base reads with substitutions, plus reads with insertions and deletions
    (the pile-up of those depends on the order of the reads), some unmappable ones"""
    from kma_amd import synth
    names, seqs = synth.make_gene_db(n_families=40, variants=5, seed=77)
    reads, *_ = synth.make_reads(seqs, 12000, read_len=150, sub_rate=0.01, random_frac=0.02, seed=78)
    rag = [r for r in reads]
    rng = np.random.default_rng(5)
    for g in rng.integers(0, len(seqs), 60):
        if len(seqs[g]) > 260:
            rag += synth.make_long_reads(seqs[g], 40, read_len=250, sub=0.01, dele=0.012, ins=0.012, seed=int(g) + 1)
    order = rng.permutation(len(rag))
    rag = [rag[i] for i in order]
    return names, seqs, rag, [b"q%d" % i for i in range(len(rag))]


def main():
    out_dir, max_frag = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("KMA_SHARE_GPU") == "1" else int(os.environ["LOCAL_RANK"])
    dist.init_process_group("gloo")
    from kma_amd import binding, formats
    from kma_amd.dist import run_se_sharded, shard_bounds
    from test_dist_gloo import _guard_collectives
    _guard_collectives()          # every collective of run_se_sharded must take tensors placed for the group's backend
    names, seqs, rag, rnames = case()
    prefix = os.path.join(out_dir, "db")
    if rank == 0:
        formats.write_index(prefix, names, seqs)
    dist.barrier()
    lo, hi = shard_bounds(len(rag), rank, world)
    db = binding.KmaHipDB(prefix, device=local)
    o = run_se_sharded(db, formats.pack_ragged(rag[lo:hi]), names=rnames[lo:hi], max_frag=max_frag, frag_path=os.path.join(out_dir, "frag.%d"))
    cons = [None] * world
    dist.all_gather_object(cons, o["consensus"])
    if rank == 0:
        tn = [x.rstrip("\n") for x in open(prefix + ".name")]
        merged = {}
        for c in cons:
            merged.update(c)
        with open(os.path.join(out_dir, "sharded.res"), "w") as f:
            for r in o["rows"]:
                t = r.template_id
                line = binding.KmaHipDB.res_line(tn[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
                if line:
                    f.write(line)
        with open(os.path.join(out_dir, "sharded.fsa"), "w") as f:
            for t in sorted(merged):
                f.write(f">{tn[t - 1]}\n{merged[t]}\n")
        with open(os.path.join(out_dir, "sharded.frag"), "wb") as f:
            for r in range(world):
                f.write(open(os.path.join(out_dir, "frag.%d" % r), "rb").read())
        np.save(os.path.join(out_dir, "owner.npy"), o["owner"])
    db.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
