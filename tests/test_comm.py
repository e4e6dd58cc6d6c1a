"""The communicator of the C-ABI on the CPU (no GPU call: bootstrap, barrier, mailboxes and the host all-to-all run over POSIX
shared memory): three processes exchange ragged blocks and must get back exactly what was meant for them."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from kma_amd import binding
    L = C.CDLL(binding.LIB_PATH)
    L.kmahip_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.kmahip_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.kmahip_comm_alltoallv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.kmahip_comm_barrier.argtypes = [C.c_void_p]
    L.kmahip_comm_destroy.argtypes = [C.c_void_p]
    L.kmahip_last_error.restype = C.c_char_p
    return L


def _block(src, dst):
    rng = np.random.default_rng(100 * src + dst)
    return rng.integers(0, 256, int(rng.integers(0, 5000)) if (src + dst) % 4 else 0, dtype=np.uint8)


def _worker(rank, world, key, q):
    try:
        L = _lib()
        c = C.c_void_p()
        assert L.kmahip_comm_init(rank, world, key.encode(), b"shm", C.byref(c)) == 0, L.kmahip_last_error()
        mine = np.array([rank * 7 + 1, rank], np.int64)
        got = np.zeros(2 * world, np.int64)
        assert L.kmahip_comm_allgather(c, mine.ctypes.data, mine.nbytes, got.ctypes.data) == 0, L.kmahip_last_error()
        assert got.tolist() == [x for r in range(world) for x in (r * 7 + 1, r)]
        for rep in range(3):                           # (several exchanges in a row: the payload files are numbered)
            blocks = [_block(rank + 10 * rep, d) for d in range(world)]
            send = np.concatenate(blocks) if sum(len(b) for b in blocks) else np.zeros(1, np.uint8)
            sb = np.array([len(b) for b in blocks], np.int64)
            allsb = np.zeros(world * world, np.int64)
            assert L.kmahip_comm_allgather(c, sb.ctypes.data, sb.nbytes, allsb.ctypes.data) == 0
            rb = np.ascontiguousarray(allsb.reshape(world, world)[:, rank])
            recv = np.zeros(max(1, int(rb.sum())), np.uint8)
            assert L.kmahip_comm_alltoallv(c, send.ctypes.data, sb.ctypes.data, recv.ctypes.data, rb.ctypes.data, 0, None) == 0, L.kmahip_last_error()
            want = [_block(s + 10 * rep, rank) for s in range(world)]
            assert np.array_equal(recv[:int(rb.sum())], np.concatenate(want) if int(rb.sum()) else np.zeros(0, np.uint8))
        assert L.kmahip_comm_barrier(c) == 0
        L.kmahip_comm_destroy(c)
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_three_ranks_exchange_over_shared_memory():
    import __graft_entry__ as ge
    ge.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"test{os.getpid()}"
    ps = [ctx.Process(target=_worker, args=(r, 3, key, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert res == [(0, "ok"), (1, "ok"), (2, "ok")], res
    assert not [f for f in os.listdir("/dev/shm") if f.startswith(f"kmahip_{key}")]      # nothing left behind


def test_one_rank_communicator_is_a_no_op():
    L = _lib()
    c = C.c_void_p()
    assert L.kmahip_comm_init(0, 1, b"solo", b"shm", C.byref(c)) == 0
    a = np.arange(5, dtype=np.int64)
    b = np.zeros(5, np.int64)
    assert L.kmahip_comm_allgather(c, a.ctypes.data, a.nbytes, b.ctypes.data) == 0 and np.array_equal(a, b)
    sb = np.array([40], np.int64)
    assert L.kmahip_comm_alltoallv(c, a.ctypes.data, sb.ctypes.data, b.ctypes.data, sb.ctypes.data, 0, None) == 0
    L.kmahip_comm_destroy(c)
