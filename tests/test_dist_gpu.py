"""The N > 1 path with the HIP kernels as compute (tests/test_dist_gloo.py covers the same logic on the CPU with the oracle):
two ranks on one card over gloo, bench.py started the way the driver starts it, and the RCCL entry of the C-ABI."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, script, *args, env=None):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script, *args]
    return subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)


def test_two_ranks_on_the_hip_path_reproduce_the_single_process_result(tmp_path):
    import oracle
    from kma_amd import binding, formats, synth
    r = _torchrun(2, os.path.join(ROOT, "tests", "dist_hip_worker.py"), str(tmp_path), env={"KMA_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    got = np.load(tmp_path / "sharded.npz")
    names, seqs = synth.make_gene_db(n_families=40, variants=5, seed=77)
    reads, *_ = synth.make_reads(seqs, 20000, read_len=150, sub_rate=0.01, random_frac=0.02, seed=78)
    batch = formats.pack_fixed(reads)
    prefix = str(tmp_path / "db")
    db = binding.KmaHipDB(prefix)
    try:
        (rc_flag, flag, T_off, T), h = db.map_se(batch)
        cc = db.conclave_se(batch.length, T_off, h)
    finally:
        db.close()
    assert np.array_equal(got["aln"], h["alignment_scores"].astype(np.int64))
    assert np.array_equal(got["uniq"], h["uniq_alignment_scores"].astype(np.int64))
    assert np.array_equal(got["w"], cc["w_scores"].astype(np.int64)) and np.array_equal(got["depth"], cc["depth"].astype(np.int64))
    assert np.array_equal(got["frags"], cc["fragment_counts"].astype(np.int64)) and np.array_equal(got["reads"], cc["read_counts"].astype(np.int64))
    assert np.array_equal(got["tmpl"], cc["tmpl"])
    # ... and the single-process HIP result is the oracle's
    odb = oracle.OracleDB(prefix)
    e = odb.scan_se(batch)
    o = odb.align_se(batch, *e)
    assert np.array_equal(o["alignment_scores"], h["alignment_scores"]) and int(got["w"].sum()) > 0


def test_bench_with_two_ranks_reports_two_gpus_and_both_scalings():
    """bench.py --gpus 2 from a plain python process: it must start its ranks itself (gloo rehearsal on one card) and print
    n_gpus as the process group saw it."""
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--reads", "400000",
                        "--steps", "2", "--warmup", "1", "--no-cpu"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["strong_scaling"]["reads_per_gpu"] == 200000 and line["strong_scaling"]["value"] > 0


RCCL_WORKER = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(rank)
from kma_amd import binding
L = binding.lib()
rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = Uid()
if rank == 0:
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
box = [C.string_at(C.byref(uid), 128)]
dist.broadcast_object_list(box, src=0)
C.memmove(C.byref(uid), box[0], 128)
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
D = 5001
a = torch.arange(D, dtype=torch.int64, device="cuda") * (rank + 1)
u = torch.full((D,), 1 << 40, dtype=torch.int64, device="cuda") + rank
L.kmahip_allreduce_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
rc = L.kmahip_allreduce_scores(comm, C.c_void_p(a.data_ptr()), C.c_void_p(u.data_ptr()), D, None)
assert rc == 0, L.kmahip_last_error()
torch.cuda.synchronize()
tot = sum(r + 1 for r in range(world))
assert torch.equal(a.cpu(), torch.arange(D, dtype=torch.int64) * tot)
assert torch.equal(u.cpu(), torch.full((D,), (1 << 40) * world + sum(range(world)), dtype=torch.int64))
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
dist.destroy_process_group()
print("rccl ok", rank, world)
'''


def test_c_abi_allreduce_goes_through_rccl(tmp_path):
    """kmahip_allreduce_scores (ncclAllReduce, ncclUint64, ncclSum) on an ncclComm_t the host program made: over two devices when
    the box has them, else over a one-rank communicator (the call still runs through librccl)."""
    import torch
    world = 2 if torch.cuda.device_count() >= 2 else 1
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    r = _torchrun(world, str(script), ROOT)
    assert r.returncode == 0 and b"rccl ok" in r.stdout, (r.stdout.decode()[-1000:], r.stderr.decode()[-3000:])


@pytest.mark.parametrize("world,max_frag", [(2, 0), (3, 1500)])
def test_sharded_whole_pipeline_reproduces_the_single_process_files(tmp_path, world, max_frag):
    """`.res` lines, consensus sequences and `.frag` rows of the read-sharded run (score vectors summed, ConClave outputs summed,
    traced reads gathered by template owner in stream order) against kmahip_run_se + kmahip_frag_write on the whole stream --
    with reads that carry insertions and deletions, and a max_frag small enough for the chunks to straddle the shards."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_pipeline_worker as W
    from kma_amd import binding, formats
    r = _torchrun(world, os.path.join(ROOT, "tests", "dist_pipeline_worker.py"), str(tmp_path), str(max_frag), env={"KMA_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    names, seqs, rag, rnames = W.case()
    prefix = str(tmp_path / "db")
    batch = formats.pack_ragged(rag)
    db = binding.KmaHipDB(prefix)
    try:
        o = db.run_se(batch, max_frag=max_frag)
        rows = db.frag_write(str(tmp_path / "single.frag"), batch, o["rc"], o["tmpl"], o["n_hits"], o["trace_stats"], rnames, max_frag=max_frag)
    finally:
        db.close()
    res = "".join(filter(None, (binding.KmaHipDB.res_line(names[x.template_id - 1], x, o["cover"][x.template_id], o["aln_len"][x.template_id],
                                                          o["depth"][x.template_id]) for x in o["rows"])))
    fsa = "".join(f">{names[t - 1]}\n{o['consensus'][t]}\n" for t in sorted(o["consensus"]))
    assert res.count("\n") > 30 and rows > 10000
    assert open(tmp_path / "sharded.res").read() == res
    assert open(tmp_path / "sharded.fsa").read() == fsa
    assert open(tmp_path / "sharded.frag", "rb").read() == open(tmp_path / "single.frag", "rb").read()
    owner = np.load(tmp_path / "owner.npy")
    assert len(set(owner.tolist())) == world          # every rank owned templates


def test_multi_rank_host_program_writes_the_single_gpu_files(tmp_path):
    """python -m kma_amd.dist_map on 2 ranks (gloo, one card) against examples/kmahip_map on the same FASTQ: .res and .fsa byte
    for byte, .frag.gz after decompression"""
    import gzip
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_pipeline_worker as W
    from kma_amd import formats, synth
    names, seqs, rag, rnames = W.case()
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, rag, prefix="q")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one"), "-1t1"], check=True, stderr=subprocess.DEVNULL)
    e = dict(os.environ, PYTHONPATH=ROOT)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "-m", "kma_amd.dist_map", "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "two"), "--backend", "gloo", "--share-gpu"]
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert open(tmp_path / "one.res").read() == open(tmp_path / "two.res").read() and open(tmp_path / "one.res").read().count("\n") > 30
    assert open(tmp_path / "one.fsa").read() == open(tmp_path / "two.fsa").read()
    assert gzip.open(tmp_path / "one.frag.gz").read() == gzip.open(tmp_path / "two.frag.gz").read()


def test_multi_rank_host_program_over_rccl_when_two_devices_are_visible(tmp_path):
    """the same program under --backend nccl (= RCCL), one rank per device: only where the box has two (the 1-GPU test boxes skip
    it; the placement of every collective's tensors is checked on the CPU in tests/test_dist_gloo.py)"""
    import gzip
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_pipeline_worker as W
    from kma_amd import formats, synth
    names, seqs, rag, rnames = W.case()
    prefix = str(tmp_path / "db")
    formats.write_index(prefix, names, seqs)
    fq = str(tmp_path / "reads.fq")
    synth.write_fastq(fq, rag, prefix="q")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run([os.path.join(ROOT, "examples", "kmahip_map"), "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "one"), "-1t1"], check=True, stderr=subprocess.DEVNULL)
    e = dict(os.environ, PYTHONPATH=ROOT)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "-m", "kma_amd.dist_map", "-i", fq, "-t_db", prefix, "-o", str(tmp_path / "two"), "--backend", "nccl"]
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    assert open(tmp_path / "one.res").read() == open(tmp_path / "two.res").read()
    assert open(tmp_path / "one.fsa").read() == open(tmp_path / "two.fsa").read()
    assert gzip.open(tmp_path / "one.frag.gz").read() == gzip.open(tmp_path / "two.frag.gz").read()
