"""fastgz.h (the Huffman-only gzip members of the .frag.gz writer) through kmahip_gzip_member: whatever goes in comes back out of
zlib's inflate, for empty, tiny, one-symbol, skewed (code lengths beyond 15 bits before limiting), text-like and random inputs."""
import ctypes as C
import gzip
import zlib

import numpy as np
import pytest

from kma_amd import binding


def _member(data: bytes) -> bytes:
    L = binding.lib()
    L.kmahip_gzip_member.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.kmahip_gzip_member.restype = C.c_int
    cap = len(data) + len(data) // 4 + 4096
    dst = C.create_string_buffer(cap)
    out = C.c_int64(0)
    assert L.kmahip_gzip_member(data, len(data), dst, cap, C.byref(out)) == 0
    return dst.raw[:out.value]


def _cases():
    rng = np.random.default_rng(7)
    yield "empty", b""
    yield "one byte", b"A"
    yield "one symbol", b"A" * 100000
    yield "two symbols", b"AB" * 5000 + b"A"
    yield "random", rng.integers(0, 256, 700001, dtype=np.uint8).tobytes()
    # Fibonacci-like counts: an unlimited Huffman code would be ~25 bits deep
    fib = [1, 1]
    while len(fib) < 26:
        fib.append(fib[-1] + fib[-2])
    skew = np.concatenate([np.full(c, 40 + i, np.uint8) for i, c in enumerate(fib)])
    rng.shuffle(skew)
    yield "skewed", skew.tobytes()
    rows = []
    for i in range(20000):
        s = "".join(rng.choice(list("ACGT"), 150))
        rows.append(f"{s}\t1\t{int(rng.integers(100, 300))}\t{int(rng.integers(0, 900))}\t{int(rng.integers(900, 1500))}\tgene_{int(rng.integers(0, 5000))}_v\tread{i} x\n")
    yield "frag rows", "".join(rows).encode()
    yield "several blocks", bytes(rng.integers(65, 70, (1 << 18) * 3 + 17, dtype=np.uint8))


@pytest.mark.parametrize("name,data", list(_cases()), ids=[n for n, _ in _cases()])
def test_member_inflates_to_the_input(name, data):
    z = _member(data)
    assert gzip.decompress(z) == data
    d = zlib.decompressobj(31)
    assert d.decompress(z) == data and d.eof and d.unused_data == b""
    # members concatenate into one gzip file
    assert gzip.decompress(z + _member(b"tail\n") + z) == data + b"tail\n" + data


def test_size_is_close_to_level_1_on_fragment_rows():
    data = dict(_cases())["frag rows"]
    z = _member(data)
    ref = zlib.compressobj(1, zlib.DEFLATED, 31)
    r = ref.compress(data) + ref.flush()
    assert len(z) < 1.3 * len(r), (len(z), len(r))


def test_every_length_around_the_checksum_fold_inflates():
    """the member's CRC-32 is folded 64 bytes at a time with carry-less multiplication, then 16 at a time, then by table: every length
    from 0 to 400 (and a few large odd ones) must come back through zlib, which checks the checksum"""
    rng = np.random.default_rng(11)
    blob = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    for n in list(range(0, 401)) + [4095, 4096, 4097, 65535, 100001, (1 << 20) - 1]:
        data = blob[7:7 + n]
        assert gzip.decompress(_member(data)) == data, n
