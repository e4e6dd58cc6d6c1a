"""The RCCL transport of the C host (kma_amd/csrc/comm.hip) on a box with ONE device: with KMAHIP_COMM_FORCE_RCCL=1 a one-rank
communicator of backend "rccl" is a real RCCL communicator (ncclGetUniqueId -> ncclCommInitRank with the 128-byte id by value,
ncclCommCount / ncclCommUserRank, a self-test at start-up) and every exchange goes through it: the SUM all-reduce of u64 vectors
(ncclUint64 from rccl.h) and the grouped ncclSend / ncclRecv all-to-all, self included. What N ranks add on top is only that the
id travels through the shared-memory mailbox, which tests/test_comm.py and test_shard_gpu.py cover with the staged backend.
Second half: the whole sharded run of examples/kmahip_map (kmahip_run_se_sharded / _pe_ / _chain_ / _mt1_sharded: both SUMs of
SURVEY 8e and the gather by template owner) over that communicator writes the files of the plain one-rank run."""
import ctypes as C
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAP = os.path.join(ROOT, "examples", "kmahip_map")


def _lib():
    from kma_amd import binding
    L = binding.lib()
    L.kmahip_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.kmahip_comm_allreduce_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.kmahip_comm_alltoallv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.kmahip_comm_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.kmahip_comm_is_rccl.argtypes = [C.c_void_p]
    L.kmahip_comm_destroy.argtypes = [C.c_void_p]
    L.kmahip_last_error.restype = C.c_char_p
    return L


def _describe(L, c):
    buf = C.create_string_buffer(512)
    L.kmahip_comm_describe(c, buf, 512)
    return dict(kv.split("=") for kv in buf.value.decode().split())


def test_one_rank_rccl_communicator_carries_the_allreduce_and_the_grouped_send_recv(monkeypatch):
    import torch
    L = _lib()
    torch.zeros(1, device="cuda:0")
    monkeypatch.setenv("KMAHIP_COMM_FORCE_RCCL", "1")
    c = C.c_void_p()
    assert L.kmahip_comm_init(0, 1, f"rccl{os.getpid()}".encode(), b"rccl", C.byref(c)) == 0, L.kmahip_last_error()
    assert L.kmahip_comm_is_rccl(c) == 1
    d = _describe(L, c)
    assert d["backend"] == "rccl" and d["rccl_nranks"] == "1" and d["rccl_rank"] == "0" and int(d["rccl_version"]) > 20000, d
    # SUM over one rank: the vector itself, every bit of the 64 (a wrong datatype enum would truncate or convert)
    rng = np.random.default_rng(3)
    h = rng.integers(0, 1 << 63, 10001, dtype=np.uint64) | (np.uint64(1) << np.uint64(63))
    v = torch.from_numpy(h.view(np.int64)).to("cuda:0")
    torch.cuda.synchronize()
    assert L.kmahip_comm_allreduce_u64(c, v.data_ptr(), v.numel(), None) == 0, L.kmahip_last_error()
    assert np.array_equal(v.cpu().numpy().view(np.uint64), h)
    # the all-to-all with itself: ncclSend + ncclRecv to the own rank inside one group, odd sizes, and an empty exchange
    for nbytes in (1, 12345, 3 << 20, 0):
        src = torch.from_numpy(rng.integers(0, 256, max(nbytes, 1), dtype=np.uint8)).to("cuda:0")
        dst = torch.zeros(max(nbytes, 1), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        sb = np.array([nbytes], np.int64)
        assert L.kmahip_comm_alltoallv(c, src.data_ptr(), sb.ctypes.data, dst.data_ptr(), sb.ctypes.data, 1, None) == 0, L.kmahip_last_error()
        assert torch.equal(src[:nbytes], dst[:nbytes])
    d = _describe(L, c)
    assert d["allreduces"] == "1" and d["alltoallvs"] == "4", d      # (the empty exchange: a group with nothing in it)
    L.kmahip_comm_destroy(c)


def test_without_the_switch_one_rank_needs_no_transport(monkeypatch):
    L = _lib()
    monkeypatch.delenv("KMAHIP_COMM_FORCE_RCCL", raising=False)
    c = C.c_void_p()
    assert L.kmahip_comm_init(0, 1, b"solo_rccl", b"rccl", C.byref(c)) == 0
    assert L.kmahip_comm_is_rccl(c) == 0 and _describe(L, c)["backend"] == "none"
    L.kmahip_comm_destroy(c)


def _run(args, env=None):
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    r = subprocess.run([MAP] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return r.stderr.decode()


def _same_files(a, b, min_rows=30):
    assert open(a + ".res", "rb").read() == open(b + ".res", "rb").read() and open(a + ".res").read().count("\n") > min_rows
    assert open(a + ".fsa", "rb").read() == open(b + ".fsa", "rb").read()
    assert open(a + ".aln", "rb").read() == open(b + ".aln", "rb").read()
    assert gzip.open(a + ".frag.gz").read() == gzip.open(b + ".frag.gz").read()


def _comm_line(err):
    m = re.search(r"# kmahip_map rank 0 comm: (.*)", err)
    assert m, err[-2000:]
    return dict(kv.split("=") for kv in m.group(1).split())


@pytest.mark.parametrize("mode", ["1t1", "chain", "pe"])
def test_sharded_run_over_a_real_rccl_communicator_writes_the_one_rank_files(tmp_path, mode):
    from test_shard_gpu import _case, _pe_case
    if mode == "pe":
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
        prefix, r1, r2 = _pe_case(tmp_path, n_pairs=6000)
        args = ["-ipe", r1, r2, "-t_db", prefix, "-1t1", "-apm", "p", "-mf", "1001"]
    else:
        prefix, fq = _case(tmp_path, n=9000)
        args = ["-i", fq, "-t_db", prefix, "-mf", "1500"] + (["-1t1"] if mode == "1t1" else [])
    _run(args + ["-o", str(tmp_path / "one")])
    err = _run(args + ["-o", str(tmp_path / "rccl")], env={"KMAHIP_COMM_FORCE_RCCL": "1", "KMAHIP_COMM": "rccl"})
    d = _comm_line(err)
    # RCCL saw the rank, both SUMs (score vectors, ConClave's outputs) and the device legs of the gather by owner went through it
    assert d["backend"] == "rccl" and d["rccl_nranks"] == "1" and int(d["allreduces"]) >= 2 and int(d["alltoallvs"]) >= 4, d
    _same_files(str(tmp_path / "one"), str(tmp_path / "rccl"))


def test_sharded_mt1_run_over_a_real_rccl_communicator(tmp_path):
    import golden_util
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    m = golden_util.load_mt1(tmp_path / "g")
    g = m["dir"]
    args = ["-i", m["fastq"], "-t_db", m["prefix"], "-Mt1", "1", "-bcNano"]
    _run(args + ["-o", str(tmp_path / "one")])
    err = _run(args + ["-o", str(tmp_path / "rccl")], env={"KMAHIP_COMM_FORCE_RCCL": "1"})
    d = _comm_line(err)
    assert d["backend"] == "rccl" and d["rccl_nranks"] == "1" and int(d["allreduces"]) >= 1 and int(d["alltoallvs"]) >= 3, d
    _same_files(str(tmp_path / "one"), str(tmp_path / "rccl"), min_rows=1)
    assert open(str(tmp_path / "rccl") + ".res", "rb").read() == open(os.path.join(g, "out.res"), "rb").read()
