"""`kma -i reads.fq -t_db db -o out -1t1` over several GPUs of one node: one process per GPU (torch.distributed.run), the reads
sharded over the ranks, `.res`, `.fsa` and `.frag.gz` written as the single-GPU program examples/kmahip_map writes them.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        -m kma_amd.dist_map -i reads.fq.gz -t_db db -o out

What crosses ranks is described in kma_amd/dist.py (run_se_sharded): two all-reduces of per-template vectors and one all-to-all
of the traced reads to the owners of their templates. Every rank parses the input itself and keeps its contiguous part of the
stream (stage 1 is host work and not sharded here)."""
import argparse
import os
import shutil

import numpy as np
import torch
import torch.distributed as dist

from . import binding
from .dist import run_se_sharded, shard_bounds
from .formats import ReadBatch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", required=True)
    ap.add_argument("-t_db", required=True)
    ap.add_argument("-o", required=True)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) or gloo")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on one card: every rank on cuda:0 (needs --backend gloo)")
    a = ap.parse_args()
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    with binding.Ingest(a.i) as ing:
        got = ing.next(1 << 62)
        ing.status()
    if got is None:
        batch, names = ReadBatch(np.zeros(1, np.uint64), np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(1, np.int64)), []
    else:
        whole, names, _ = got
        lo, hi = shard_bounds(whole.n, rank, world)
        so, no = whole.seq_off, whole.N_off
        batch = ReadBatch(whole.seq[so[lo]:so[hi]].copy(), (so[lo:hi + 1] - so[lo]).copy(), whole.length[lo:hi].copy(),
                          whole.N[no[lo]:no[hi]].copy(), (no[lo:hi + 1] - no[lo]).copy())
        names = names[lo:hi]
        del whole
    db = binding.KmaHipDB(a.t_db, device=local)
    frag = a.o + ".frag.%d.gz"
    o = run_se_sharded(db, batch, names=names, frag_path=frag)
    cons = [None] * world
    dist.all_gather_object(cons, o["consensus"])      # (pickled; torch stages it on the current device under nccl)
    if rank == 0:
        tn = [x.rstrip("\n") for x in open(a.t_db + ".name")]
        merged = {}
        for c in cons:
            merged.update(c)
        # what examples/kmahip_map.c writes: a row and a consensus entry per significant template the reference prints a row for
        with open(a.o + ".res", "w") as f, open(a.o + ".fsa", "w") as g:
            f.write("#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n")
            for r in o["rows"]:
                t = r.template_id
                if not r.significant:
                    continue
                line = binding.KmaHipDB.res_line(tn[t - 1], r, o["cover"][t], o["aln_len"][t], o["depth"][t])
                if not line:
                    continue
                f.write(line)
                s = merged.get(t, "").replace("-", "")               # printConsensus (printconsensus.c:38-60)
                g.write(f">{tn[t - 1]}\n")
                for x in range(0, len(s), 60):
                    g.write(s[x:x + 60] + "\n")
        # gzip members concatenate: the owners hold contiguous template ranges, so rank order is template order
        with open(a.o + ".frag.gz", "wb") as f:
            for r in range(world):
                with open(frag % r, "rb") as g:
                    shutil.copyfileobj(g, f)
                os.unlink(frag % r)
    db.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
