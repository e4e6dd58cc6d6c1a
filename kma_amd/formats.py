"""Binary stream / index formats of the KMA mapping path.

Pure numpy helpers used by the tests, the benchmark and the Python binding:
  * 2-bit read packing into the batch layout the C-ABI takes (compdna.c:99-127)
  * S1 / S2 stream record parsers (runinput.c:765-787, ankers.c:30-50)
  * .comp.b / .length.b / .seq.b readers and a writer of the same format
    (hashmapkma.c:275-455, :722-775) for synthetic benchmark databases.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np


# ----------------------------------------------------------------------------
# read batches
# ----------------------------------------------------------------------------
@dataclass
class ReadBatch:
    """CSR batch of 2-bit packed reads, the layout `kmahip_reads` points into.

    seq      u64 words, every read followed by ONE zero pad word
    seq_off  i64[n+1] word offset of read i (pad included in the stride)
    length   i32[n]
    N        i32   concatenated sorted N positions
    N_off    i64[n+1]
    """
    seq: np.ndarray
    seq_off: np.ndarray
    length: np.ndarray
    N: np.ndarray
    N_off: np.ndarray

    @property
    def n(self):
        return len(self.length)


def pack_fixed(reads: np.ndarray) -> ReadBatch:
    """Pack an [n, L] uint8 code matrix (0..3, 4 = N)."""
    n, L = reads.shape
    words = (L + 31) // 32
    pad = words * 32 - L
    r = reads
    isn = r == 4
    r2 = np.where(isn, 0, r).astype(np.uint64)
    if pad:
        r2 = np.concatenate([r2, np.zeros((n, pad), np.uint64)], axis=1)
    r2 = r2.reshape(n, words, 32)
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    w = (r2 << shifts[None, None, :]).sum(axis=2, dtype=np.uint64)
    seq = np.zeros((n, words + 1), np.uint64)
    seq[:, :words] = w
    seq_off = np.arange(n + 1, dtype=np.int64) * (words + 1)
    rows, cols = np.nonzero(isn)
    N = cols.astype(np.int32)
    cnt = np.bincount(rows, minlength=n)
    N_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return ReadBatch(seq.reshape(-1), seq_off, np.full(n, L, np.int32), N, N_off)


def pack_ragged(reads) -> ReadBatch:
    """Pack a list of uint8 code arrays of differing length."""
    n = len(reads)
    lens = np.array([len(r) for r in reads], np.int32)
    words = (lens.astype(np.int64) + 31) // 32
    seq_off = np.concatenate([[0], np.cumsum(words + 1)]).astype(np.int64)
    seq = np.zeros(int(seq_off[-1]), np.uint64)
    Ns, cnt = [], np.zeros(n, np.int64)
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    for i, r in enumerate(reads):
        L = len(r)
        if L == 0:
            continue
        isn = r == 4
        r2 = np.where(isn, 0, r).astype(np.uint64)
        W = int(words[i])
        if W * 32 != L:
            r2 = np.concatenate([r2, np.zeros(W * 32 - L, np.uint64)])
        seq[seq_off[i]:seq_off[i] + W] = (r2.reshape(W, 32) << shifts[None, :]).sum(axis=1, dtype=np.uint64)
        p = np.nonzero(isn)[0]
        cnt[i] = len(p)
        Ns.append(p.astype(np.int32))
    N = np.concatenate(Ns) if Ns else np.zeros(0, np.int32)
    N_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    return ReadBatch(seq, seq_off, lens, N.astype(np.int32), N_off)


def unpack_words(words: np.ndarray, length: int) -> np.ndarray:
    """2-bit words -> uint8 codes (N positions come back as 0)."""
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    b = ((words[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return b[:length]


# ----------------------------------------------------------------------------
# stream records
# ----------------------------------------------------------------------------
def parse_s1(buf: bytes):
    """S1 records -> list of dict(seqlen, seq(u64), N(i32), hdr(bytes), pair)."""
    out, o = [], 0
    while o + 16 <= len(buf):
        seqlen, complen, nN, hdrlen = struct.unpack_from("<4i", buf, o)
        o += 16
        seq = np.frombuffer(buf, np.uint64, complen, o); o += 8 * complen
        N = np.frombuffer(buf, np.int32, nN, o); o += 4 * nN
        hl = abs(hdrlen)
        hdr = buf[o:o + hl]; o += hl
        out.append(dict(seqlen=seqlen, seq=seq, N=N, hdr=hdr, pair=hdrlen < 0))
    return out


def parse_s2(buf: bytes):
    """S2 records (ankers.c:30-50) up to the `-nReads` terminator."""
    out, o = [], 0
    while o + 4 <= len(buf):
        (first,) = struct.unpack_from("<i", buf, o)
        if first < 0:
            return out, -first
        seqlen, complen, nN, rc_flag, nT, hdrlen, flag = struct.unpack_from("<7i", buf, o)
        o += 28
        seq = np.frombuffer(buf, np.uint64, complen, o); o += 8 * complen
        N = np.frombuffer(buf, np.int32, nN, o); o += 4 * nN
        T = np.frombuffer(buf, np.int32, nT, o); o += 4 * nT
        hdr = buf[o:o + hdrlen]; o += hdrlen
        out.append(dict(seqlen=seqlen, seq=seq, N=N, rc_flag=rc_flag, T=T, hdr=hdr, flag=flag))
    return out, None


# ----------------------------------------------------------------------------
# index files
# ----------------------------------------------------------------------------
@dataclass
class CompDB:
    DB_size: int
    mlen: int
    prefix_len: int
    prefix: int
    size: int          # number of buckets (power of two)
    n: int
    v_index: int
    null_index: int
    exist: np.ndarray
    values: np.ndarray
    key_index: np.ndarray
    value_index: np.ndarray
    kmersize: int
    flag: int


def read_lengths(prefix) -> np.ndarray:
    """<prefix>.length.b -> template_lengths[DB_size] (int32; entry 0 = the index's k-mer count slot)."""
    raw = np.fromfile(prefix + ".length.b", dtype=np.int32)
    return raw[1:1 + int(raw[0])].copy()


def read_comp_b(path) -> CompDB:
    b = open(path, "rb").read()
    DB_size, mlen, prefix_len = struct.unpack_from("<3I", b, 0)
    prefix, size, n, v_index, null_index = struct.unpack_from("<5Q", b, 12)
    o = 52
    if size - 1 == (1 << (2 * mlen)) - 1:
        raise ValueError("megamap index not supported")
    edt = np.uint32 if n <= 0xFFFFFFFF else np.uint64
    exist = np.frombuffer(b, edt, size, o); o += exist.nbytes
    vdt = np.uint16 if DB_size < 65535 else np.uint32
    values = np.frombuffer(b, vdt, v_index, o); o += values.nbytes
    kdt = np.uint32 if mlen <= 16 else np.uint64
    key_index = np.frombuffer(b, kdt, n + 1, o); o += key_index.nbytes
    idt = np.uint32 if v_index < 0xFFFFFFFF else np.uint64
    value_index = np.frombuffer(b, idt, n, o); o += value_index.nbytes
    if o + 8 <= len(b):
        kmersize, flag = struct.unpack_from("<2I", b, o)
    else:
        kmersize, flag = mlen, 0
    return CompDB(DB_size, mlen, prefix_len, prefix, size, n, v_index, null_index,
                  exist, values, key_index, value_index, kmersize, flag)


def comp_db_mapping(db: CompDB):
    """{kmer: tuple(template ids)} -- for semantic comparison of two indexes."""
    out = {}
    vals = db.values
    for k, vi in zip(db.key_index[:db.n].tolist(), db.value_index.tolist()):
        c = int(vals[vi])
        out[k] = tuple(vals[vi + 1: vi + 1 + c].tolist())
    return out


def _all_kmers(seqs, k):
    """Forward-strand k-mers (as u64) and their 1-based template ids."""
    ks, ts = [], []
    for t, s in enumerate(seqs, start=1):
        L = len(s)
        if L < k:
            continue
        x = s.astype(np.uint64)
        # rolling via cumulative base-4 windows
        km = np.zeros(L - k + 1, np.uint64)
        for i in range(k):
            km = (km << np.uint64(2)) | x[i:L - k + 1 + i]
        ks.append(km)
        ts.append(np.full(len(km), t, np.uint32))
    return np.concatenate(ks), np.concatenate(ts)


def write_index(prefix, names, seqs, k=16):
    """Write <prefix>.comp.b/.length.b/.seq.b/.name for N-free templates.

    Same file format as `kma index` (App. A); the bucket count, key order
    inside a bucket and value-list order differ from the reference's builder
    but the k-mer -> template-set mapping and the full de-duplication of equal
    sets (compress.c:218) are the same, which is all the mapping path reads.
    """
    assert 4 <= k <= 16
    n_t = len(seqs)
    DB_size = n_t + 1
    km, tid = _all_kmers(seqs, k)
    order = np.lexsort((tid, km))
    km, tid = km[order], tid[order]
    # unique (kmer, template) pairs
    keep = np.ones(len(km), bool)
    keep[1:] = (km[1:] != km[:-1]) | (tid[1:] != tid[:-1])
    km, tid = km[keep], tid[keep]
    # group by kmer
    starts = np.nonzero(np.concatenate([[True], km[1:] != km[:-1]]))[0]
    ends = np.concatenate([starts[1:], [len(km)]])
    ukm = km[starts]
    n = len(ukm)
    cnt = (ends - starts).astype(np.int64)
    # hash each template list to dedupe equal sets
    P = np.uint64(0x9E3779B97F4A7C15)
    h = (tid.astype(np.uint64) + np.uint64(1)) * P
    h ^= h >> np.uint64(29)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    csum = np.concatenate([np.zeros(1, np.uint64), np.cumsum(h, dtype=np.uint64)])
    cx = np.concatenate([np.zeros(1, np.uint64), np.bitwise_xor.accumulate(h * np.uint64(0x94D049BB133111EB))])
    sig = np.stack([csum[ends] - csum[starts], cx[ends] ^ cx[starts], cnt.astype(np.uint64)], axis=1)
    # exact dedupe: group by signature then verify by content
    sig_view = np.ascontiguousarray(sig).view([("a", np.uint64), ("b", np.uint64), ("c", np.uint64)]).reshape(-1)
    _, first_idx, inv = np.unique(sig_view, return_index=True, return_inverse=True)
    # verify no signature collision (content equality with representative)
    rep = first_idx[inv]
    maxc = int(cnt.max())
    for j in range(maxc):
        m = cnt > j
        if not np.array_equal(tid[starts[m] + j], tid[starts[rep[m]] + j]):
            raise RuntimeError("value-set signature collision")
    vdt = np.uint16 if DB_size < 65535 else np.uint32
    # lay out unique lists in order of first appearance
    uniq_order = np.argsort(first_idx, kind="stable")
    ucnt = cnt[first_idx[uniq_order]]
    uoff = np.concatenate([[0], np.cumsum(ucnt + 1)])
    v_index = int(uoff[-1])
    values = np.zeros(v_index, vdt)
    values[uoff[:-1]] = ucnt.astype(vdt)
    # fill elements
    src_start = starts[first_idx[uniq_order]]
    rep_len = ucnt
    dst = np.repeat(uoff[:-1] + 1, rep_len) + (np.arange(int(rep_len.sum())) - np.repeat(np.cumsum(rep_len) - rep_len, rep_len))
    src = np.repeat(src_start, rep_len) + (np.arange(int(rep_len.sum())) - np.repeat(np.cumsum(rep_len) - rep_len, rep_len))
    values[dst] = tid[src].astype(vdt)
    rank = np.empty(len(uniq_order), np.int64)
    rank[uniq_order] = np.arange(len(uniq_order))
    vi_of_key = uoff[:-1][rank[inv]]
    # buckets
    size = 1 << 20
    while size < n:
        size <<= 1
    bucket = (ukm & np.uint64(size - 1)).astype(np.int64)
    o2 = np.argsort(bucket, kind="stable")
    bsorted = bucket[o2]
    key_index = np.zeros(n + 1, np.uint32)
    key_index[:n] = ukm[o2].astype(np.uint32)
    value_index = vi_of_key[o2].astype(np.uint32)
    exist = np.full(size, n, np.uint32)
    firsts = np.nonzero(np.concatenate([[True], bsorted[1:] != bsorted[:-1]]))[0]
    exist[bsorted[firsts]] = firsts.astype(np.uint32)
    # sentinel key must belong to a different bucket than the last run
    last_b = int(bsorted[-1])
    key_index[n] = np.uint32((last_b + 1) & (size - 1))
    with open(prefix + ".comp.b", "wb") as f:
        f.write(struct.pack("<3I5Q", DB_size, k, 0, 0, size, n, v_index, n))
        f.write(exist.tobytes()); f.write(values.tobytes())
        f.write(key_index.tobytes()); f.write(value_index.tobytes())
        f.write(struct.pack("<2I", k, 0))
    lens = np.zeros(DB_size, np.int32)
    lens[0] = k
    lens[1:] = [len(s) for s in seqs]
    with open(prefix + ".length.b", "wb") as f:
        f.write(struct.pack("<i", DB_size)); f.write(lens.tobytes())
    shifts = (np.uint64(62) - np.uint64(2) * np.arange(32, dtype=np.uint64))
    with open(prefix + ".seq.b", "wb") as f:
        for s in seqs:
            W = (len(s) >> 5) + 1
            x = np.zeros(W * 32, np.uint64); x[:len(s)] = s
            f.write((x.reshape(W, 32) << shifts[None, :]).sum(axis=1, dtype=np.uint64).tobytes())
    with open(prefix + ".name", "w") as f:
        for nm in names:
            f.write(nm + "\n")
