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
