"""The C-ABI library builds, loads and exports every symbol include/kmahip.h declares (no GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "kmahip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kmahip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_all_declared_symbols():
    import __graft_entry__ as ge
    ge.build()
    from kma_amd import binding
    lib = ctypes.CDLL(binding.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"libkmahip.so does not export {s}"


def test_default_params_match_reference_defaults():
    # kma.c:327-336,1307-1328
    from kma_amd import binding
    p = binding.default_params()
    assert (p.rw.M, p.rw.MM, p.rw.U, p.rw.W1, p.rw.Wl, p.rw.Mn, p.rw.PE) == (1, -2, -1, -3, -6, 0, 7)
    d = [[p.rw.d[i][j] for j in range(5)] for i in range(5)]
    assert d[0] == [1, -2, -2, -2, 0] and d[4] == [0, 0, 0, 0, 0] and d[2][2] == 1
    assert (p.minlen, p.mq, p.scoreT, p.mrc, p.minFrac) == (16, 0, 0.5, 0.0, 1.0)


def test_header_and_example_are_plain_c99():
    """The boundary is a C ABI: the header and the example host program must compile as pedantic C99 (no GPU, no link)."""
    import subprocess
    for src in ("kmahip_s2.c", "kmahip_res.c"):
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                               "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", src)])


def test_sparse_index_is_rejected_loudly(tmp_path):
    """`kma index -Sparse` output (prefix_len / prefix in the .comp.b header) must not be taken for a full index: the reference
    maps against it with save_kmers_sparse and without stage 3 (kma.c:1499-1501). The check precedes every device call."""
    import shutil
    import struct
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_util
    from kma_amd import binding
    os.makedirs(tmp_path / "g")
    g = golden_util.load_se(tmp_path / "g")
    L = binding.lib()
    ref = os.path.join(ROOT, "oracle", "_ref", "kma")
    prefixes = []
    if os.path.exists(ref):
        fsa = tmp_path / "db.fsa"
        import gzip
        fsa.write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "se", "db.fsa.gz")).read())
        subprocess.check_call([ref, "index", "-i", str(fsa), "-o", str(tmp_path / "sp"), "-Sparse", "TG"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        prefixes.append(str(tmp_path / "sp"))
    # and a header patched by hand (prefix_len = 2, prefix = 0b1110), for boxes without the reference binary
    for ext in (".comp.b", ".length.b", ".seq.b", ".name"):
        shutil.copy(g["prefix"] + ext, str(tmp_path / "patched") + ext)
    with open(str(tmp_path / "patched") + ".comp.b", "r+b") as f:
        f.seek(8)
        f.write(struct.pack("<IQ", 2, 14))
    prefixes.append(str(tmp_path / "patched"))
    for p in prefixes:
        h = ctypes.c_void_p()
        rc = L.kmahip_db_open(p.encode(), ctypes.byref(h))
        assert rc == -3 and h.value is None, (p, rc)
        assert b"sparse" in L.kmahip_last_error()


def test_chain_unpinned_reads_counts_what_the_reference_reads_from_stale_memory():
    """kmahip_chain_unpinned_reads (host arrays, no GPU): an N among the first k - 1 bases AND a longer read before it in the stream"""
    import numpy as np
    from kma_amd import binding
    lib = ctypes.CDLL(binding.LIB_PATH)
    lib.kmahip_chain_unpinned_reads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)]
    length = np.array([100, 150, 120, 150, 90, 200], np.int32)
    # read 0: N at 3 but nothing longer before; read 2: N at 5 behind the 150; read 3: N at 20 (not among the first 15); read 4: N at 0
    Ns = np.array([3, 5, 20, 0, 40], np.int32)
    N_off = np.array([0, 1, 1, 2, 3, 5, 5], np.int64)
    longest, count = ctypes.c_int32(), ctypes.c_int64()
    assert lib.kmahip_chain_unpinned_reads(length.ctypes.data, Ns.ctypes.data, N_off.ctypes.data, 6, 16, 0, ctypes.byref(longest), ctypes.byref(count)) == 0
    assert (count.value, longest.value) == (2, 200)
    # the same batch behind one that held a 300-base read: read 0 counts too
    assert lib.kmahip_chain_unpinned_reads(length.ctypes.data, Ns.ctypes.data, N_off.ctypes.data, 6, 16, 300, ctypes.byref(longest), ctypes.byref(count)) == 0
    assert (count.value, longest.value) == (3, 300)
