"""ctypes view of oracle/libkma_oracle.so -- the CPU checker (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libkma_oracle.so")
REF_KMA = os.path.join(ORACLE_DIR, "_ref", "kma")


class Rewards(C.Structure):
    _fields_ = [("M", C.c_int), ("MM", C.c_int), ("U", C.c_int), ("W1", C.c_int),
                ("Wl", C.c_int), ("Mn", C.c_int), ("PE", C.c_int), ("d", (C.c_int * 5) * 5)]


class AlignParams(C.Structure):
    _fields_ = [("minlen", C.c_int), ("mq", C.c_int), ("scoreT", C.c_double), ("mrc", C.c_double), ("minFrac", C.c_double)]


class ChainRec(C.Structure):
    _fields_ = [("rc_flag", C.c_int), ("emit_rc", C.c_int), ("nT", C.c_int), ("q_start", C.c_int), ("q_end", C.c_int), ("T", C.c_void_p)]


class PeRec(C.Structure):
    _fields_ = [("present", C.c_int), ("mate", C.c_int), ("rc", C.c_int), ("rc_flag", C.c_int), ("flag", C.c_int),
                ("nT", C.c_int), ("T", C.POINTER(C.c_int))]


class PeOut(C.Structure):
    _fields_ = [("kind", C.c_int), ("swapped", C.c_int), ("n_hits", C.c_int), ("best", C.c_int), ("best_r", C.c_int),
                ("flagA", C.c_int), ("flagB", C.c_int), ("rcA", C.c_int), ("rcB", C.c_int), ("n_hits_r", C.c_int),
                ("tmpl", C.c_void_p), ("score", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p)]


def _build():
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build()
        L = C.CDLL(LIB)
        L.orc_db_load.restype = C.c_void_p
        L.orc_db_load.argtypes = [C.c_char_p]
        L.orc_db_free.argtypes = [C.c_void_p]
        L.orc_default_rewards.argtypes = [C.POINTER(Rewards)]
        L.orc_hash_get.restype = C.c_int64
        L.orc_hash_get.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_scan_se_batch.restype = C.c_int64
        L.orc_scan_se_batch.argtypes = [C.c_void_p, C.POINTER(Rewards), C.c_int, C.c_int64] + [C.c_void_p] * 9 + [C.c_int64]
        L.orc_align_se_batch.restype = C.c_int64
        L.orc_align_se_batch.argtypes = [C.c_void_p, C.POINTER(Rewards), C.POINTER(AlignParams), C.c_int64] + [C.c_void_p] * 18
        L.orc_nw_tap.restype = None
        L.orc_nw_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.POINTER(Rewards), C.c_void_p]
        L.orc_scan_pe.restype = C.c_int
        L.orc_scan_pe.argtypes = [C.c_void_p, C.POINTER(Rewards), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(PeRec * 2), C.c_void_p, C.c_void_p]
        L.orc_aligner_new.restype = C.c_void_p
        L.orc_aligner_new.argtypes = [C.c_void_p]
        L.orc_aligner_free.argtypes = [C.c_void_p]
        L.orc_align_pe.restype = C.c_int
        L.orc_align_pe.argtypes = [C.c_void_p, C.POINTER(Rewards), C.POINTER(AlignParams),
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.POINTER(PeOut), C.c_void_p, C.c_void_p]
        L.orc_rc.restype = None
        L.orc_rc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleDB:
    def __init__(self, prefix):
        self.prefix = prefix
        self.h = lib().orc_db_load(prefix.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load index {prefix}")
        self.rw = Rewards()
        lib().orc_default_rewards(C.byref(self.rw))

    def close(self):
        if self.h:
            lib().orc_db_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def hash_get(self, key):
        return lib().orc_hash_get(self.h, int(key))

    def scan_se(self, batch, exhaustive=0, t_cap=None):
        """-> rc_flag[n], flag[n], T_off[n+1], T[...] (numpy)."""
        n = batch.n
        rc_flag = np.zeros(n, np.int32)
        flag = np.zeros(n, np.int32)
        T_off = np.zeros(n + 1, np.int64)
        cap = t_cap or max(1024, 64 * n)
        while True:
            T = np.zeros(cap, np.int32)
            seq = np.ascontiguousarray(batch.seq)
            Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32))
            r = lib().orc_scan_se_batch(self.h, C.byref(self.rw), exhaustive, n, _p(seq), _p(batch.seq_off),
                                        _p(batch.length), _p(Nn), _p(batch.N_off),
                                        _p(rc_flag), _p(flag), _p(T_off), _p(T), cap)
            if r >= 0:
                return rc_flag, flag, T_off, T[:r]
            cap = -r + 16

    def scan_chain(self, batch, minlen=16, coverT=0.1, mrs=0.5, exhaustive=0):
        """default template finder (no -1t1), read by read -> list per read of (rc_flag, emit_rc, q_start, q_end, T array)"""
        L = lib()
        L.orc_scan_chain.restype = C.c_int
        L.orc_scan_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        out = []
        rec = (ChainRec * 64)()
        pool = np.zeros(1 << 20, np.int32)
        seq = np.ascontiguousarray(batch.seq)
        for i in range(batch.n):
            Ln = int(batch.length[i])
            words = np.zeros(((Ln + 31) >> 5) + 2, np.uint64)
            w = seq[batch.seq_off[i]:batch.seq_off[i] + ((Ln + 31) >> 5)]
            words[:len(w)] = w
            Ni = batch.N[batch.N_off[i]:batch.N_off[i + 1]]
            Nl = np.zeros(len(Ni) + 2, np.int32)
            Nl[0] = len(Ni)
            Nl[1:1 + len(Ni)] = Ni
            r = L.orc_scan_chain(self.h, C.byref(self.rw), exhaustive, minlen, coverT, mrs, _p(words), Ln, _p(Nl), rec, 64, _p(pool), len(pool))
            assert r >= 0, "oracle chain finder: capacity"
            base = pool.ctypes.data
            out.append([(rec[x].rc_flag, rec[x].emit_rc, rec[x].q_start, rec[x].q_end,
                         pool[(rec[x].T - base) // 4:(rec[x].T - base) // 4 + rec[x].nT].copy()) for x in range(r)])
        return out

    def align_se(self, batch, rc_flag, flag, T_off, T, minlen=16, mq=0, scoreT=0.5, mrc=0.0):
        """-> dict(n_hits, best_score, out_flag, tmpl, start, end, score (CSR at T_off), alignment_scores, uniq)"""
        n = batch.n
        ap = AlignParams(minlen, mq, scoreT, mrc, 1.0)
        cap = max(1, len(T))
        out = dict(n_hits=np.zeros(n, np.int32), best_score=np.zeros(n, np.int32), out_flag=np.zeros(n, np.int32),
                   tmpl=np.zeros(cap, np.int32), start=np.zeros(cap, np.int32), end=np.zeros(cap, np.int32),
                   score=np.zeros(cap, np.int32))
        import struct
        dbsize = struct.unpack("<I", open(self.prefix + ".comp.b", "rb").read(4))[0]
        out["alignment_scores"] = np.zeros(dbsize, np.uint64)
        out["uniq_alignment_scores"] = np.zeros(dbsize, np.uint64)
        seq = np.ascontiguousarray(batch.seq)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32))
        Tn = np.ascontiguousarray(T if len(T) else np.zeros(1, np.int32), np.int32)
        lib().orc_align_se_batch(self.h, C.byref(self.rw), C.byref(ap), n, _p(seq), _p(batch.seq_off), _p(batch.length),
                                 _p(Nn), _p(batch.N_off), _p(np.ascontiguousarray(rc_flag, np.int32)),
                                 _p(np.ascontiguousarray(flag, np.int32)), _p(np.ascontiguousarray(T_off, np.int64)), _p(Tn),
                                 _p(out["n_hits"]), _p(out["best_score"]), _p(out["out_flag"]), _p(out["tmpl"]),
                                 _p(out["start"]), _p(out["end"]), _p(out["score"]),
                                 _p(out["alignment_scores"]), _p(out["uniq_alignment_scores"]))
        return out

    def scan_pe(self, seq1, len1, N1, seq2, len2, N2, exhaustive=0, union=False, force=False):
        """One pair (padded u64 word arrays + N position arrays) -> list of up to two record dicts in stream order.
        union: the union pairing (-apm u, and `-ipe` without -apm) instead of the pairing penalty (-apm p)"""
        import struct
        D = struct.unpack("<I", open(self.prefix + ".comp.b", "rb").read(4))[0]
        T1 = np.zeros(2 * D + 8, np.int32); T2 = np.zeros(2 * D + 8, np.int32)
        recs = (PeRec * 2)()
        s1 = np.ascontiguousarray(np.concatenate([seq1, np.zeros(2, np.uint64)]))
        s2 = np.ascontiguousarray(np.concatenate([seq2, np.zeros(2, np.uint64)]))
        n1 = np.ascontiguousarray(N1 if len(N1) else np.zeros(1, np.int32), np.int32)
        n2 = np.ascontiguousarray(N2 if len(N2) else np.zeros(1, np.int32), np.int32)
        fn = lib().orc_scan_pe_force if force else (lib().orc_scan_pe_union if union else lib().orc_scan_pe)          # (force: stage 2 of -apm f)
        fn.restype = C.c_int
        fn.argtypes = lib().orc_scan_pe.argtypes
        ret = fn(self.h, C.byref(self.rw), exhaustive, _p(s1), len1, _p(n1), len(N1),
                                _p(s2), len2, _p(n2), len(N2), C.byref(recs), _p(T1), _p(T2))
        out = []
        for r in recs:
            if r.present:
                out.append(dict(mate=r.mate, rc=r.rc, rc_flag=r.rc_flag, flag=r.flag,
                                T=np.array([r.T[i] for i in range(r.nT)], np.int32)))
        return ret, out


def conclave(n_hits, read_score, q_len, q_len2, off, tmpl, start, end, alignment_scores, uniq, tlen):
    """Stage 3b on arrays -> dict(tmpl, start, end per record; w_scores, fragmentCounts, readCounts, depth per template)."""
    n = len(n_hits)
    D = len(tlen)
    i32 = lambda x: np.ascontiguousarray(x, np.int32)
    pad = lambda x: i32(x if len(x) else np.zeros(1, np.int32))
    out = dict(tmpl=np.zeros(n, np.int32), start=np.zeros(n, np.int32), end=np.zeros(n, np.int32),
               w_scores=np.zeros(D, np.uint64), fragmentCounts=np.zeros(D, np.uint32), readCounts=np.zeros(D, np.uint32),
               depth=np.zeros(D, np.uint64))
    L = lib()
    L.orc_conclave.restype = C.c_int
    L.orc_conclave.argtypes = [C.c_int64] + [C.c_void_p] * 18
    a = [i32(n_hits), i32(read_score), i32(q_len), i32(q_len2), np.ascontiguousarray(off, np.int64), pad(tmpl), pad(start),
         pad(end), np.ascontiguousarray(alignment_scores, np.uint64), np.ascontiguousarray(uniq, np.uint64), i32(tlen)]
    rc = L.orc_conclave(n, *[_p(x) for x in a], _p(out["tmpl"]), _p(out["start"]), _p(out["end"]), _p(out["w_scores"]),
                        _p(out["fragmentCounts"]), _p(out["readCounts"]), _p(out["depth"]))
    assert rc == 0, "ConClave found no template for a multi-hit record"
    return out


def res_stats(w_scores, tlen, evalue=0.05, scoreT=0.5):
    """-> dict(expected (as printed), q_value, p_value, significant) per template (runkma.c:765-783)."""
    D = len(tlen)
    out = dict(expected=np.zeros(D), q_value=np.zeros(D), p_value=np.zeros(D), significant=np.zeros(D, np.int32))
    L = lib()
    L.orc_res_stats.restype = C.c_int
    L.orc_res_stats.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double] + [C.c_void_p] * 4
    L.orc_res_stats(D, _p(np.ascontiguousarray(w_scores, np.uint64)), _p(np.ascontiguousarray(tlen, np.int32)), evalue, scoreT,
                    _p(out["expected"]), _p(out["q_value"]), _p(out["p_value"]), _p(out["significant"]))
    return out


class Assembly:
    """Pile-up + consensus of one template (oracle/assembly.c)."""

    def __init__(self, odb, t, t_len):
        L = lib()
        L.orc_assembly_new.restype = C.c_void_p
        L.orc_assembly_new.argtypes = [C.c_int]
        L.orc_assembly_free.argtypes = [C.c_void_p]
        L.orc_assembly_add.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_assembly_call.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_char_p]
        L.orc_db_template.restype = C.c_void_p
        L.orc_db_template.argtypes = [C.c_void_p, C.c_int]
        self.h = L.orc_assembly_new(int(t_len))
        self.tseq = L.orc_db_template(odb.h, int(t))
        self.t_len = int(t_len)
        self.n = 0

    def add(self, trace, read):
        """trace = OracleAligner.align_trace result; read = the oriented read it was computed from"""
        rd = np.ascontiguousarray(read[trace["clip_start"]:], np.uint8)
        lib().orc_assembly_add(self.h, trace["cols"], len(trace["cols"]), _p(rd), trace["start"], trace["score"])
        self.n += 1

    def call(self, bcd=1, evalue=0.05, caller=0, sig=0):
        """caller 1 = nanoCaller, sig 1 = significantAnd90Nuc (both: -bcNano)"""
        out = np.zeros(4, np.int64)
        cons = C.create_string_buffer(int(2 * self.t_len + 64 + 2 * 1024 * 1024))
        L = lib()
        L.orc_assembly_call2.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_char_p]
        L.orc_assembly_call2(self.h, self.tseq, bcd, evalue, caller, sig, _p(out), cons)
        return dict(cover=int(out[0]), aln_len=int(out[1]), depth=int(out[2]), asm_len=int(out[3]), consensus=cons.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_assembly_free(self.h)
            self.h = None


def res_identity_columns(call, t_len):
    """runkma.c:792-809: the five consensus columns of a `.res` row as printed, or None when the row is not written"""
    if call["cover"] <= 0:
        return None
    import ctypes
    ident = 100.0 * call["cover"] / t_len
    cover = 100.0 * call["aln_len"] / t_len
    q_id = 100.0 * call["cover"] / call["aln_len"]
    q_cover = 100.0 * t_len / call["aln_len"]
    depth = float(np.longdouble(call["depth"]) / np.longdouble(t_len))
    if not (1.0 <= ident and 0 < ident):
        return None
    return tuple("%.2f" % x for x in (ident, cover, q_id, q_cover, depth))


def cigar_of(cols, clip_start=0, clip_end=0):
    """makeCigar (sam.c:30-98): run-length code of the column classes, soft clips around it"""
    out = [f"{clip_start}S"] if clip_start else []
    i = 0
    while i < len(cols):
        j = i
        while j < len(cols) and cols[j] == cols[i]:
            j += 1
        out.append(f"{j - i}{cols[i]}")
        i = j
    if clip_end:
        out.append(f"{clip_end}S")
    return "".join(out)


def rc_packed(seq, length, N):
    """compdna.c:228-256 on numpy arrays -> (rc words, rc N positions)."""
    words = (length + 31) // 32
    s = np.ascontiguousarray(np.concatenate([seq[:words], np.zeros(2, np.uint64)]))
    fN = np.ascontiguousarray(np.concatenate([[len(N)], N]).astype(np.int32))
    rs = np.zeros(words + 2, np.uint64)
    rN = np.zeros(len(N) + 2, np.int32)
    lib().orc_rc(_p(s), length, _p(fN), _p(rs), _p(rN))
    return rs[:words], rN[1:1 + len(N)]


class OracleAligner:
    """Stateful aligner (per-template indexes are built lazily and kept)."""

    def __init__(self, odb, minlen=16, mq=0, scoreT=0.5, mrc=0.0):
        import struct
        self.odb = odb
        self.h = lib().orc_aligner_new(odb.h)
        self.ap = AlignParams(minlen, mq, scoreT, mrc, 1.0)
        D = struct.unpack("<I", open(odb.prefix + ".comp.b", "rb").read(4))[0]
        self.alignment_scores = np.zeros(D, np.uint64)
        self.uniq_alignment_scores = np.zeros(D, np.uint64)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_aligner_free(self.h)
            self.h = None

    def align_trace(self, read, t, minlen=None):
        """Stage 3c for one read (uint8 codes 0-4, oriented like the template) -> None if dropped, else dict(score, start,
        end, aln_len, clip_start, clip_end, match, tGaps, qGaps, mapQ, cigar)"""
        rd = np.ascontiguousarray(read, np.uint8)
        stats = np.zeros(10, np.int32)
        cap = 4 * len(rd) + 4 * 4096
        cols = C.create_string_buffer(cap)
        L = lib()
        L.orc_align_trace.restype = C.c_int
        L.orc_align_trace.argtypes = [C.c_void_p, C.POINTER(Rewards), C.POINTER(AlignParams), C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p, C.c_char_p, C.c_int]
        n = L.orc_align_trace(self.h, C.byref(self.odb.rw), C.byref(self.ap), _p(rd), len(rd), int(t), _p(stats), cols, cap)
        assert n >= 0
        if n == 0:
            return None
        keys = ("score", "start", "end", "aln_len", "clip_start", "clip_end", "match", "tGaps", "qGaps", "mapQ")
        out = dict(zip(keys, (int(x) for x in stats)))
        out["cols"] = cols.raw[:n]
        out["cigar"] = cigar_of(out["cols"].decode(), out["clip_start"], out["clip_end"])
        return out

    def align_trace_mt1(self, read, t, one2one=0, exhaustive=0):
        """One raw read of a `-Mt1 t` run: anker_rc (strand) + KMA() + read filter -> (result dict as align_trace or None,
        is_rc, the read as it was aligned)"""
        rd = np.ascontiguousarray(read, np.uint8).copy()
        stats = np.zeros(10, np.int32)
        cap = 4 * len(rd) + 4 * 4096
        cols = C.create_string_buffer(cap)
        is_rc = C.c_int()
        L = lib()
        L.orc_align_trace_mt1.restype = C.c_int
        L.orc_align_trace_mt1.argtypes = [C.c_void_p, C.POINTER(Rewards), C.POINTER(AlignParams), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        n = L.orc_align_trace_mt1(self.h, C.byref(self.odb.rw), C.byref(self.ap), _p(rd), len(rd), int(t), int(one2one), int(exhaustive),
                                  _p(stats), cols, cap, C.byref(is_rc))
        assert n >= 0
        if n == 0:
            return None, is_rc.value, rd
        keys = ("score", "start", "end", "aln_len", "clip_start", "clip_end", "match", "tGaps", "qGaps", "mapQ")
        out = dict(zip(keys, (int(x) for x in stats)))
        out["cols"] = cols.raw[:n]
        out["cigar"] = cigar_of(out["cols"].decode(), out["clip_start"], out["clip_end"])
        return out, is_rc.value, rd

    def align_pe(self, seqA, lenA, NA, flagA, seqB, lenB, NB, flagB, T):
        nT = len(T)
        arr = [np.zeros(nT + 2, np.int32) for _ in range(4)]
        out = PeOut()
        out.tmpl, out.score, out.start, out.end = (a.ctypes.data for a in arr)
        pad = lambda s: np.ascontiguousarray(np.concatenate([s, np.zeros(2, np.uint64)]))
        nn = lambda N: np.ascontiguousarray(N if len(N) else np.zeros(1, np.int32), np.int32)
        sA, sB, nA, nB = pad(seqA), pad(seqB), nn(NA), nn(NB)
        Tn = np.ascontiguousarray(T, np.int32)
        ret = lib().orc_align_pe(self.h, C.byref(self.odb.rw), C.byref(self.ap), _p(sA), lenA, _p(nA), len(NA), flagA,
                                 _p(sB), lenB, _p(nB), len(NB), flagB, _p(Tn), nT, C.byref(out),
                                 _p(self.alignment_scores), _p(self.uniq_alignment_scores))
        n = out.n_hits
        return ret, dict(kind=out.kind, swapped=out.swapped, n_hits=n, best=out.best, best_r=out.best_r,
                         flagA=out.flagA, flagB=out.flagB, n_hits_r=out.n_hits_r,
                         tmpl=arr[0][:n + out.n_hits_r].copy(), score=arr[1][:n + out.n_hits_r].copy(),
                         start=arr[2][:n + out.n_hits_r].copy(), end=arr[3][:n + out.n_hits_r].copy())
