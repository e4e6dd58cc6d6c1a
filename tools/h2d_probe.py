#!/usr/bin/env python3
"""hipMemcpy of a large PAGEABLE host array to the device: in one call, and in pieces -- what the uploads of kmahip_run_* can expect.
usage (GPU box): python3 tools/h2d_probe.py [GB]"""
import ctypes as C
import sys
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
n = int(gb * (1 << 30))
src = np.ones(n, np.uint8)
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0
p = src.ctypes.data


def whole():
    assert hip.hipMemcpy(d, C.c_void_p(p), C.c_size_t(n), 1) == 0


def pieces(mb):
    def f():
        ch = mb << 20
        for o in range(0, n, ch):
            assert hip.hipMemcpy(C.c_void_p(d.value + o), C.c_void_p(p + o), C.c_size_t(min(ch, n - o)), 1) == 0
    return f


for label, fn in (("whole", whole), ("64 MB pieces", pieces(64)), ("16 MB pieces", pieces(16)), ("whole", whole), ("256 MB pieces", pieces(256))):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        fn()
        hip.hipDeviceSynchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{label:14s}", " ".join(f"{1e3 * t:7.1f} ms" for t in ts), f"  best {gb / min(ts):5.1f} GB/s")
# a fresh array each time (pages never touched by a copy before)
for label, fn_of in (("whole, fresh", lambda q: (lambda: hip.hipMemcpy(d, C.c_void_p(q), C.c_size_t(n), 1))),):
    ts = []
    for _ in range(3):
        a = np.ones(n, np.uint8)
        t0 = time.perf_counter()
        fn_of(a.ctypes.data)()
        hip.hipDeviceSynchronize()
        ts.append(time.perf_counter() - t0)
        del a
    print(f"{label:14s}", " ".join(f"{1e3 * t:7.1f} ms" for t in ts))
