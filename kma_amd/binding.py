"""ctypes binding of libkmahip.so (include/kmahip.h) -- the only way Python
reaches the HIP path.  There is no CPU fallback: if the shared library is not
built (python -c 'import __graft_entry__ as g; g.build()') import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMAHIP_LIB") or os.path.join(_HERE, "libkmahip.so")


class Rewards(C.Structure):
    _fields_ = [("M", C.c_int32), ("MM", C.c_int32), ("U", C.c_int32), ("W1", C.c_int32),
                ("Wl", C.c_int32), ("Mn", C.c_int32), ("PE", C.c_int32), ("d", (C.c_int32 * 5) * 5)]


class Params(C.Structure):
    _fields_ = [("rw", Rewards), ("exhaustive", C.c_int32), ("minlen", C.c_int32), ("mq", C.c_int32),
                ("scoreT", C.c_double), ("mrc", C.c_double), ("minFrac", C.c_double), ("ts", C.c_int32), ("apm", C.c_int32)]


class DBInfo(C.Structure):
    _fields_ = [("DB_size", C.c_uint32), ("kmersize", C.c_uint32), ("n_kmers", C.c_uint64),
                ("n_values", C.c_uint64), ("hash_bytes", C.c_uint64), ("total_bytes", C.c_uint64),
                ("tseq_words", C.c_uint64)]


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("seq", C.c_void_p), ("seq_off", C.c_void_p), ("len", C.c_void_p),
                ("N", C.c_void_p), ("N_off", C.c_void_p), ("seq_words", C.c_int64), ("N_total", C.c_int64),
                ("max_len", C.c_int32), ("q_start", C.c_void_p), ("q_end", C.c_void_p)]


class Cands(C.Structure):
    _fields_ = [("rc_flag", C.c_void_p), ("flag", C.c_void_p), ("T_off", C.c_void_p), ("T", C.c_void_p),
                ("T_cap", C.c_int64)]


class Hits(C.Structure):
    _fields_ = [("n_hits", C.c_void_p), ("best_score", C.c_void_p), ("flag", C.c_void_p), ("tmpl", C.c_void_p),
                ("score", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p),
                ("alignment_scores", C.c_void_p), ("uniq_alignment_scores", C.c_void_p), ("rc", C.c_void_p)]


class PeRecs(C.Structure):
    _fields_ = [("mate", C.c_void_p), ("rc", C.c_void_p), ("rc_flag", C.c_void_p), ("flag", C.c_void_p),
                ("R_off", C.c_void_p), ("T", C.c_void_p), ("T_cap", C.c_int64)]


class Conclave(C.Structure):
    _fields_ = [("tmpl", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p), ("w_scores", C.c_void_p),
                ("fragment_counts", C.c_void_p), ("read_counts", C.c_void_p), ("depth", C.c_void_p)]


class ResRow(C.Structure):
    _fields_ = [("template_id", C.c_int32), ("template_length", C.c_int32), ("score", C.c_uint64), ("expected", C.c_uint32),
                ("significant", C.c_int32), ("q_value", C.c_double), ("p_value", C.c_double)]


class Traces(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("ops_off", C.c_void_p), ("n_ops", C.c_void_p), ("ops", C.c_void_p), ("ops_cap", C.c_int64)]


class Assembly(C.Structure):
    _fields_ = [("cover", C.c_void_p), ("aln_len", C.c_void_p), ("depth", C.c_void_p), ("asm_len", C.c_void_p),
                ("consensus", C.c_void_p), ("consensus_off", C.c_void_p), ("consensus_cap", C.c_int64), ("consensus_used", C.c_int64)]


class Trim(C.Structure):
    _fields_ = [("min_phred", C.c_int32), ("min_q", C.c_int32), ("hardmask_q", C.c_int32), ("min_len", C.c_int32),
                ("max_len", C.c_int32)]


class ReadBatchC(C.Structure):
    _fields_ = [("reads", Reads), ("names", C.c_void_p), ("name_off", C.c_void_p), ("pair", C.c_void_p), ("records", C.c_int64)]


class ChainParams(C.Structure):
    _fields_ = [("minlen", C.c_int32), ("pad_", C.c_int32), ("coverT", C.c_double), ("mrs", C.c_double)]


class ChainRecs(C.Structure):
    _fields_ = [("rec_cap", C.c_int64), ("T_cap", C.c_int64), ("n_recs", C.c_int64), ("n_T", C.c_int64), ("read", C.c_void_p),
                ("rc_flag", C.c_void_p), ("emit_rc", C.c_void_p), ("q_start", C.c_void_p), ("q_end", C.c_void_p), ("T_off", C.c_void_p),
                ("T", C.c_void_p)]


class AssembleOpts(C.Structure):
    _fields_ = [("max_frag", C.c_int64), ("evalue", C.c_double), ("bcd", C.c_int32), ("order", C.c_int32), ("caller", C.c_int32), ("sig90", C.c_int32),
                ("frag_rank", C.c_void_p), ("support", C.c_double)]


class Run(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("rows_cap", C.c_int64), ("n_rows", C.c_int64), ("assembly", Assembly),
                ("tmpl", C.c_void_p), ("n_hits", C.c_void_p), ("rc", C.c_void_p), ("trace_stats", C.c_void_p), ("ms", C.c_double * 6),
                ("caller", C.c_int32), ("sig90", C.c_int32), ("support", C.c_double)]


class ScanStats(C.Structure):
    _fields_ = [("probes", C.c_uint64), ("value_elems", C.c_uint64), ("active_strands", C.c_uint64),
                ("hash_probes", C.c_uint64), ("prefilter_probes", C.c_uint64)]


class AlignStats(C.Structure):
    _fields_ = [("lookups", C.c_uint64), ("mem_bases", C.c_uint64), ("dp_cells", C.c_uint64), ("tasks", C.c_uint64)]


class TraceStats(C.Structure):
    _fields_ = [("problems", C.c_uint64), ("dp_cells", C.c_uint64), ("mems", C.c_uint64), ("reads", C.c_uint64)]


class KmaHipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if "KMAHIP_LIB" not in os.environ:
            # build on demand, also when a source is newer than the library (the .so is git-ignored but travels to the GPU box:
            # a stale one must never be what gets tested or benchmarked). `make -q` costs milliseconds when everything is current;
            # there is no other implementation to fall back to
            import shutil
            import subprocess
            csrc = os.path.join(_HERE, "csrc")
            stale = not os.path.exists(LIB_PATH)
            if not stale and shutil.which("make"):
                stale = subprocess.call(["make", "-q", "-C", csrc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0
            if stale:
                # several ranks import this at the same moment (bench.py --gpus N, the dist tests): one builds, the others wait
                # for the lock and find the library current
                import fcntl
                with open(os.path.join(csrc, ".build.lock"), "w") as lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    try:
                        if not os.path.exists(LIB_PATH) or subprocess.call(["make", "-q", "-C", csrc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0:
                            subprocess.check_call(["make", "-C", csrc, "-j4"], stdout=subprocess.DEVNULL)
                    except (OSError, subprocess.CalledProcessError) as e:
                        raise KmaHipError(f"{LIB_PATH} is missing or older than its sources and building it failed ({e}): run __graft_entry__.build()")
                    finally:
                        fcntl.flock(lock, fcntl.LOCK_UN)
        if not os.path.exists(LIB_PATH):
            raise KmaHipError(f"{LIB_PATH} is not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIB_PATH)
        L.kmahip_last_error.restype = C.c_char_p
        L.kmahip_default_params.argtypes = [C.POINTER(Params)]
        L.kmahip_default_params.restype = None
        L.kmahip_init.argtypes = [C.c_int]
        L.kmahip_db_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.kmahip_db_close.argtypes = [C.c_void_p]
        L.kmahip_db_close.restype = None
        L.kmahip_db_get_info.argtypes = [C.c_void_p, C.POINTER(DBInfo)]
        L.kmahip_ws_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.kmahip_ws_destroy.argtypes = [C.c_void_p]
        L.kmahip_ws_destroy.restype = None
        L.kmahip_scan_se.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(Cands)]
        L.kmahip_scan_se_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(Cands), C.c_void_p]
        L.kmahip_ws_status.argtypes = [C.c_void_p, C.c_void_p]
        L.kmahip_scan_set_stats.argtypes = [C.c_void_p, C.c_int]
        L.kmahip_scan_get_stats.argtypes = [C.c_void_p, C.POINTER(ScanStats), C.c_void_p]
        L.kmahip_ws_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.kmahip_ws_get_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.kmahip_align_se_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Cands), C.POINTER(Params),
                                          C.POINTER(Hits), C.c_void_p]
        L.kmahip_align_get_stats.argtypes = [C.c_void_p, C.POINTER(AlignStats), C.c_void_p]
        L.kmahip_scan_pe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(PeRecs)]
        L.kmahip_scan_pe_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(PeRecs), C.c_void_p]
        L.kmahip_align_pe_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(PeRecs), C.POINTER(Params), C.POINTER(Hits),
                                          C.c_void_p, C.c_void_p]
        L.kmahip_scan_pe_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(PeRecs), C.c_void_p]
        L.kmahip_map_pe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(PeRecs),
                                    C.POINTER(Hits), C.c_void_p]
        L.kmahip_map_se.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(Cands), C.POINTER(Hits)]
        L.kmahip_conclave_se.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Cands), C.POINTER(Hits), C.POINTER(Conclave)]
        L.kmahip_conclave_pe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(PeRecs), C.POINTER(Hits), C.c_void_p,
                                         C.POINTER(Conclave)]
        L.kmahip_conclave_se_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Cands), C.POINTER(Hits),
                                             C.POINTER(Conclave), C.c_void_p]
        L.kmahip_conclave_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Hits),
                                              C.POINTER(Conclave)]
        L.kmahip_align_trace.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Params),
                                         C.POINTER(Traces), C.POINTER(C.c_int64)]
        L.kmahip_align_trace_mt1.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.c_int32, C.c_int, C.POINTER(Params), C.POINTER(Traces),
                                             C.c_void_p, C.POINTER(C.c_int64)]
        L.kmahip_assemble.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.POINTER(Traces), C.c_int64,
                                      C.c_int, C.c_double, C.POINTER(Assembly)]
        L.kmahip_res_line.argtypes = [C.c_char_p, C.POINTER(ResRow), C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                      C.c_char_p, C.c_int64]
        L.kmahip_res_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.kmahip_frag_write.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.kmahip_run_se.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.c_double, C.c_int, C.c_int64, C.POINTER(Run)]
        L.kmahip_run_pe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ReadBatchC), C.POINTER(Params), C.c_double, C.c_int, C.c_int64, C.c_char_p,
                                    C.POINTER(Run)]
        L.kmahip_run_mt1.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.c_int32, C.c_int, C.POINTER(Params), C.POINTER(AssembleOpts), C.POINTER(Run)]
        L.kmahip_ws_set_pe_chain.argtypes = [C.c_void_p, C.POINTER(ChainParams)]
        L.kmahip_assemble2.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.POINTER(Traces), C.POINTER(AssembleOpts),
                                       C.POINTER(Assembly)]
        L.kmahip_frag_write3.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int, C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.kmahip_frag_write2.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(Reads), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.kmahip_trace_get_stats.argtypes = [C.c_void_p, C.POINTER(TraceStats)]
        L.kmahip_trim_default.argtypes = [C.POINTER(Trim)]
        L.kmahip_trim_default.restype = None
        L.kmahip_ingest_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Trim), C.POINTER(C.c_void_p)]
        L.kmahip_ingest_next.argtypes = [C.c_void_p, C.c_int64, C.POINTER(ReadBatchC)]
        L.kmahip_ingest_phred_scale.argtypes = [C.c_void_p]
        L.kmahip_ingest_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.kmahip_ingest_counts.restype = None
        L.kmahip_ingest_close.argtypes = [C.c_void_p]
        L.kmahip_ingest_status.argtypes = [C.c_void_p]
        L.kmahip_ingest_set_batch_bases.argtypes = [C.c_void_p, C.c_int64]
        L.kmahip_ingest_close.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise KmaHipError(f"kmahip error {rc}: {lib().kmahip_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cigar_from_runs(runs, clip_start=0, clip_end=0):
    """SAM CIGAR (makeCigar, sam.c:30-98) of one read's runs"""
    out = [f"{clip_start}S"] if clip_start else []
    out += [f"{int(x) >> 2}{'=XID'[int(x) & 3]}" for x in runs]
    if clip_end:
        out.append(f"{clip_end}S")
    return "".join(out)


def index_build(fasta_paths, out_prefix, k=16):
    """kmahip_index_build: the four index files from FASTA file(s) (needs a GPU: the k-mers are sorted on the device)"""
    paths = [os.fsencode(p) for p in ([fasta_paths] if isinstance(fasta_paths, (str, bytes)) else fasta_paths)]
    arr = (C.c_char_p * len(paths))(*paths)
    L = lib()
    L.kmahip_index_build.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int]
    L.kmahip_index_build.restype = C.c_int
    _check(L.kmahip_index_build(arr, len(paths), os.fsencode(out_prefix), int(k)))


class Ingest:
    """Stage 1 on the host (kmahip_ingest_*): FASTQ / FASTA(.gz) -> trimmed, packed formats.ReadBatch batches, the
    records the reference's run_input / run_input_PE write into the S1 stream, in the same order."""

    def __init__(self, path1, path2=None, min_phred=20, min_q=0, hardmask_q=0, min_len=16, max_len=2**31 - 1, interleaved=False):
        L = lib()
        t = Trim(min_phred, min_q, hardmask_q, min_len, max_len)
        self._h = C.c_void_p()
        if interleaved:          # `-int file`: the records of one file two at a time (run_input_INT)
            if path2:
                raise ValueError("interleaved input is one file")
            L.kmahip_ingest_open_interleaved.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(C.c_void_p)]
            _check(L.kmahip_ingest_open_interleaved(os.fsencode(path1), C.byref(t), C.byref(self._h)))
            return
        _check(L.kmahip_ingest_open(os.fsencode(path1), os.fsencode(path2) if path2 else None, C.byref(t), C.byref(self._h)))

    @property
    def phred_scale(self):
        return lib().kmahip_ingest_phred_scale(self._h)

    def counts(self):
        a, b = C.c_int64(), C.c_int64()
        lib().kmahip_ingest_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def next(self, max_records=1 << 20):
        """-> (ReadBatch, names list[bytes], pair u8[n]) or None at the end of the input (copies out of the reader's buffers)"""
        from . import formats
        b = ReadBatchC()
        _check(lib().kmahip_ingest_next(self._h, max_records, C.byref(b)))
        n = b.reads.n_reads
        if n == 0:
            return None

        def arr(ptr, dt, m):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dt)), shape=(m,)).copy() if m else np.zeros(0, dt)
        batch = formats.ReadBatch(seq=arr(b.reads.seq, C.c_uint64, b.reads.seq_words), seq_off=arr(b.reads.seq_off, C.c_int64, n + 1),
                                  length=arr(b.reads.len, C.c_int32, n), N=arr(b.reads.N, C.c_int32, max(1, b.reads.N_total)),
                                  N_off=arr(b.reads.N_off, C.c_int64, n + 1))
        noff = arr(b.name_off, C.c_int64, n + 1)
        raw = C.string_at(b.names, int(noff[-1]))
        names = [raw[noff[i]:noff[i + 1] - 1] for i in range(n)]
        return batch, names, arr(b.pair, C.c_uint8, n)

    def set_batch_bases(self, max_bases):
        """a batch of next() also closes once it holds about max_bases bases (0: no such bound)"""
        _check(lib().kmahip_ingest_set_batch_bases(self._h, int(max_bases)))

    def status(self):
        """raises when the input broke off behind the records delivered so far (for callers that take everything as one batch)"""
        _check(lib().kmahip_ingest_status(self._h))

    def close(self):
        if self._h:
            lib().kmahip_ingest_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def default_params() -> Params:
    p = Params()
    lib().kmahip_default_params(C.byref(p))
    return p


class KmaHipDB:
    """An index resident in HBM + one workspace."""

    def __init__(self, prefix: str, device: int = 0):
        L = lib()
        _check(L.kmahip_init(device))
        self.prefix = prefix
        self.h = C.c_void_p()
        _check(L.kmahip_db_open(prefix.encode(), C.byref(self.h)))
        self.ws = C.c_void_p()
        _check(L.kmahip_ws_create(self.h, C.byref(self.ws)))
        self.info = DBInfo()
        _check(L.kmahip_db_get_info(self.h, C.byref(self.info)))
        self.params = default_params()

    def close(self):
        if getattr(self, "ws", None):
            lib().kmahip_ws_destroy(self.ws)
            self.ws = None
        if getattr(self, "h", None):
            lib().kmahip_db_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host buffers in / out ------------------------------------------------
    def scan_se(self, batch, exhaustive=0, t_cap=None):
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        rc_flag = np.zeros(max(n, 1), np.int32)
        flag = np.zeros(max(n, 1), np.int32)
        T_off = np.zeros(n + 1, np.int64)
        cap = t_cap or max(1024, 8 * n)
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        for _ in range(4):
            T = np.zeros(cap, np.int32)
            out = Cands(_p(rc_flag), _p(flag), _p(T_off), _p(T), cap)
            rc = lib().kmahip_scan_se(self.h, self.ws, C.byref(r), C.byref(p), C.byref(out))
            if rc == -6:  # KMAHIP_EOVERFLOW
                cap = max(cap * 2, int(T_off[n]) + 16)
                continue
            _check(rc)
            return rc_flag[:n], flag[:n], T_off, T[:T_off[n]]
        raise KmaHipError("scan_se: output capacity kept overflowing")

    # -- device resident (torch tensors) -----------------------------------------
    def scan_chain(self, batch, minlen=16, coverT=0.1, mrs=0.5, exhaustive=0):
        """Stage 2 of the default mode (kmahip_scan_chain, no -1t1) -> dict(read, rc_flag, emit_rc, q_start, q_end, T_off, T):
        one entry per S2 record, in stream order"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        cp = ChainParams(int(minlen), 0, float(coverT), float(mrs))
        L = lib()
        L.kmahip_scan_chain.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Reads), C.POINTER(Params), C.POINTER(ChainParams), C.POINTER(ChainRecs)]
        L.kmahip_scan_chain.restype = C.c_int
        rec_cap, T_cap = 2 * n + 1024, 16 * n + 4096
        for _ in range(4):
            o = dict(read=np.zeros(rec_cap, np.int64), rc_flag=np.zeros(rec_cap, np.int32), emit_rc=np.zeros(rec_cap, np.int32),
                     q_start=np.zeros(rec_cap, np.int32), q_end=np.zeros(rec_cap, np.int32), T_off=np.zeros(rec_cap + 1, np.int64),
                     T=np.zeros(T_cap, np.int32))
            out = ChainRecs(rec_cap, T_cap, 0, 0, _p(o["read"]), _p(o["rc_flag"]), _p(o["emit_rc"]), _p(o["q_start"]), _p(o["q_end"]),
                            _p(o["T_off"]), _p(o["T"]))
            rc = L.kmahip_scan_chain(self.h, self.ws, C.byref(r), C.byref(p), C.byref(cp), C.byref(out))
            if rc == -6 and (out.n_recs > rec_cap or out.n_T > T_cap):
                rec_cap, T_cap = max(rec_cap, out.n_recs + 16), max(T_cap, out.n_T + 16)
                continue
            _check(rc)
            m = out.n_recs
            for key in ("read", "rc_flag", "emit_rc", "q_start", "q_end"):
                o[key] = o[key][:m]
            o["T_off"] = o["T_off"][:m + 1]
            o["T"] = o["T"][:out.n_T]
            return o
        raise KmaHipError("scan_chain: output capacity kept overflowing")

    def scan_se_dev(self, seq, seq_off, length, N, N_off, rc_flag, flag, T_off, T, exhaustive=0, stream=None):
        """All arguments are CUDA(HIP) torch tensors; asynchronous on `stream`."""
        n = length.numel()
        r = Reads(n, seq.data_ptr(), seq_off.data_ptr(), length.data_ptr(), N.data_ptr(), N_off.data_ptr(),
                  seq.numel(), N.numel(), 0)
        out = Cands(rc_flag.data_ptr(), flag.data_ptr(), T_off.data_ptr(), T.data_ptr(), T.numel())
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        _check(lib().kmahip_scan_se_dev(self.h, self.ws, C.byref(r), C.byref(p), C.byref(out), C.c_void_p(stream or 0)))

    def status(self, stream=None):
        _check(lib().kmahip_ws_status(self.ws, C.c_void_p(stream or 0)))

    def set_stats(self, on: bool):
        _check(lib().kmahip_scan_set_stats(self.ws, int(on)))

    def get_stats(self, stream=None) -> ScanStats:
        st = ScanStats()
        _check(lib().kmahip_scan_get_stats(self.ws, C.byref(st), C.c_void_p(stream or 0)))
        return st

    def get_align_stats(self, stream=None) -> AlignStats:
        st = AlignStats()
        _check(lib().kmahip_align_get_stats(self.ws, C.byref(st), C.c_void_p(stream or 0)))
        return st

    def get_trace_stats(self) -> TraceStats:
        st = TraceStats()
        _check(lib().kmahip_trace_get_stats(self.ws, C.byref(st)))
        return st

    def set_timing(self, on: bool):
        _check(lib().kmahip_ws_set_timing(self.ws, int(on)))

    def get_timing(self, kernel=0):
        """kernel 0 = scan_se_kernel, 1 = align_tasks_kernel, 2 = scan_prefilter_kernel, 3 = seed_tasks_kernel -> (summed ms, launches)"""
        ms, n = C.c_double(), C.c_int64()
        _check(lib().kmahip_ws_get_timing(self.ws, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- stages 2 + 3a, host buffers ------------------------------------------
    def map_se(self, batch, exhaustive=0, t_cap=None):
        """-> (rc_flag, flag, T_off, T), hits dict (n_hits, best_score, flag, tmpl, score, start, end,
        alignment_scores, uniq_alignment_scores)"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        rc_flag = np.zeros(max(n, 1), np.int32)
        flag = np.zeros(max(n, 1), np.int32)
        T_off = np.zeros(n + 1, np.int64)
        cap = t_cap or max(1024, 8 * n)
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        D = int(self.info.DB_size)
        for _ in range(4):
            T = np.zeros(cap, np.int32)
            h = dict(n_hits=np.zeros(max(n, 1), np.int32), best_score=np.zeros(max(n, 1), np.int32),
                     flag=np.zeros(max(n, 1), np.int32), tmpl=np.zeros(cap, np.int32), score=np.zeros(cap, np.int32),
                     start=np.zeros(cap, np.int32), end=np.zeros(cap, np.int32),
                     alignment_scores=np.zeros(D, np.uint64), uniq_alignment_scores=np.zeros(D, np.uint64),
                     rc=np.zeros(max(n, 1), np.int32))
            out = Cands(_p(rc_flag), _p(flag), _p(T_off), _p(T), cap)
            hs = Hits(_p(h["n_hits"]), _p(h["best_score"]), _p(h["flag"]), _p(h["tmpl"]), _p(h["score"]), _p(h["start"]),
                      _p(h["end"]), _p(h["alignment_scores"]), _p(h["uniq_alignment_scores"]), _p(h["rc"]))
            rc = lib().kmahip_map_se(self.h, self.ws, C.byref(r), C.byref(p), C.byref(out), C.byref(hs))
            if rc == -6 and int(T_off[n]) > cap:
                cap = max(cap * 2, int(T_off[n]) + 16)
                continue
            _check(rc)
            for key in ("n_hits", "best_score", "flag", "rc"):
                h[key] = h[key][:n]
            return (rc_flag[:n], flag[:n], T_off, T[:T_off[n]]), h
        raise KmaHipError("map_se: output capacity kept overflowing")

    def align_se_dev(self, seq, seq_off, length, N, N_off, max_len, rc_flag, flag, T_off, T,
                     n_hits, best_score, out_flag, h_tmpl, h_score, h_start, h_end, aln_scores, uniq_scores, stream=None):
        """Stage 3a on device tensors (outputs of scan_se_dev); asynchronous on `stream`."""
        n = length.numel()
        r = Reads(n, seq.data_ptr(), seq_off.data_ptr(), length.data_ptr(), N.data_ptr(), N_off.data_ptr(),
                  seq.numel(), N.numel(), int(max_len))
        c = Cands(rc_flag.data_ptr(), flag.data_ptr(), T_off.data_ptr(), T.data_ptr(), T.numel())
        h = Hits(n_hits.data_ptr(), best_score.data_ptr(), out_flag.data_ptr(), h_tmpl.data_ptr(), h_score.data_ptr(),
                 h_start.data_ptr(), h_end.data_ptr(), aln_scores.data_ptr(), uniq_scores.data_ptr(), None)
        p = Params.from_buffer_copy(self.params)
        _check(lib().kmahip_align_se_dev(self.h, self.ws, C.byref(r), C.byref(c), C.byref(p), C.byref(h),
                                         C.c_void_p(stream or 0)))

    def scan_pe_dev(self, seq, seq_off, length, N, N_off, mate, rc, rc_flag, flag, R_off, T, exhaustive=0, stream=None):
        """Stage 2 for interleaved mates (`-apm p`) on device tensors; asynchronous on `stream`. mate / rc / rc_flag / flag: i32[2 pairs],
        R_off i64[2 pairs + 1], T i32[capacity]."""
        n = length.numel()
        r = Reads(n, seq.data_ptr(), seq_off.data_ptr(), length.data_ptr(), N.data_ptr(), N_off.data_ptr(), seq.numel(), N.numel(), 0)
        out = PeRecs(mate.data_ptr(), rc.data_ptr(), rc_flag.data_ptr(), flag.data_ptr(), R_off.data_ptr(), T.data_ptr(), T.numel())
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        _check(lib().kmahip_scan_pe_dev(self.h, self.ws, C.byref(r), C.byref(p), C.byref(out), C.c_void_p(stream or 0)))

    def align_pe_dev(self, seq, seq_off, length, N, N_off, max_len, mate, rc, rc_flag, flag, R_off, T,
                     n_hits, best_score, out_flag, h_tmpl, h_score, h_start, h_end, aln_scores, uniq_scores, out_rc, kind, stream=None):
        """Stage 3a for the records of scan_pe_dev on device tensors; asynchronous on `stream`. kind: i32[pairs]."""
        n = length.numel()
        r = Reads(n, seq.data_ptr(), seq_off.data_ptr(), length.data_ptr(), N.data_ptr(), N_off.data_ptr(), seq.numel(), N.numel(), int(max_len))
        recs = PeRecs(mate.data_ptr(), rc.data_ptr(), rc_flag.data_ptr(), flag.data_ptr(), R_off.data_ptr(), T.data_ptr(), T.numel())
        h = Hits(n_hits.data_ptr(), best_score.data_ptr(), out_flag.data_ptr(), h_tmpl.data_ptr(), h_score.data_ptr(),
                 h_start.data_ptr(), h_end.data_ptr(), aln_scores.data_ptr(), uniq_scores.data_ptr(), out_rc.data_ptr())
        p = Params.from_buffer_copy(self.params)
        _check(lib().kmahip_align_pe_dev(self.h, self.ws, C.byref(r), C.byref(recs), C.byref(p), C.byref(h), C.c_void_p(kind.data_ptr()),
                                         C.c_void_p(stream or 0)))

    # -- stage 3b (ConClave) -----------------------------------------------------------
    def _conclave_out(self, n):
        D = int(self.info.DB_size)
        o = dict(tmpl=np.zeros(max(n, 1), np.int32), start=np.zeros(max(n, 1), np.int32), end=np.zeros(max(n, 1), np.int32),
                 w_scores=np.zeros(D, np.uint64), fragment_counts=np.zeros(D, np.uint32), read_counts=np.zeros(D, np.uint32),
                 depth=np.zeros(D, np.uint64))
        return o, Conclave(_p(o["tmpl"]), _p(o["start"]), _p(o["end"]), _p(o["w_scores"]), _p(o["fragment_counts"]),
                           _p(o["read_counts"]), _p(o["depth"]))

    @staticmethod
    def _hits_struct(h):
        c = lambda a, t: np.ascontiguousarray(a if len(a) else np.zeros(1, t), t)
        keep = [c(h["n_hits"], np.int32), c(h["best_score"], np.int32), c(h["tmpl"], np.int32), c(h["start"], np.int32),
                c(h["end"], np.int32), c(h["alignment_scores"], np.uint64), c(h["uniq_alignment_scores"], np.uint64)]
        return keep, Hits(_p(keep[0]), _p(keep[1]), None, _p(keep[2]), None, _p(keep[3]), _p(keep[4]), _p(keep[5]), _p(keep[6]), None)

    def conclave_se(self, length, T_off, hits):
        """Stage 3b for the result of map_se (host arrays; `hits` may carry vectors summed over ranks) -> dict(tmpl, start,
        end per read; w_scores, fragment_counts, read_counts, depth per template)"""
        n = len(length)
        ln = np.ascontiguousarray(length, np.int32)
        to = np.ascontiguousarray(T_off, np.int64)
        r = Reads(n, None, None, _p(ln), None, None, 0, 0, 0)
        c = Cands(None, None, _p(to), None, 0)
        keep, hs = self._hits_struct(hits)
        o, oc = self._conclave_out(n)
        _check(lib().kmahip_conclave_se(self.h, self.ws, C.byref(r), C.byref(c), C.byref(hs), C.byref(oc)))
        for key in ("tmpl", "start", "end"):
            o[key] = o[key][:n]
        return o

    def conclave_pe(self, length, mate, R_off, hits):
        """Stage 3b for the result of map_pe (record slots; hits["kind"] per pair)"""
        n = len(length)
        ln = np.ascontiguousarray(length, np.int32)
        ro = np.ascontiguousarray(R_off, np.int64)
        mt = np.ascontiguousarray(mate if n else np.zeros(1, np.int32), np.int32)
        kd = np.ascontiguousarray(hits["kind"], np.int32)
        r = Reads(n, None, None, _p(ln), None, None, 0, 0, 0)
        pr = PeRecs(_p(mt), None, None, None, _p(ro), None, 0)
        keep, hs = self._hits_struct(hits)
        o, oc = self._conclave_out(n)
        _check(lib().kmahip_conclave_pe(self.h, self.ws, C.byref(r), C.byref(pr), C.byref(hs), _p(kd), C.byref(oc)))
        for key in ("tmpl", "start", "end"):
            o[key] = o[key][:n]
        return o

    def conclave_records(self, n_hits, score, q_len, q_len2, off, tmpl, start, end, alignment_scores, uniq_alignment_scores):
        """Stage 3b over explicit frag_raw records in stream order (score < 0: pair record)"""
        n = len(n_hits)
        ql = np.ascontiguousarray(q_len if n else np.zeros(1, np.int32), np.int32)
        ql2 = np.ascontiguousarray(q_len2 if n else np.zeros(1, np.int32), np.int32)
        of = np.ascontiguousarray(off, np.int64)
        assert len(of) == n + 1
        keep, hs = self._hits_struct(dict(n_hits=n_hits, best_score=score, tmpl=tmpl, start=start, end=end,
                                          alignment_scores=alignment_scores, uniq_alignment_scores=uniq_alignment_scores))
        o, oc = self._conclave_out(n)
        _check(lib().kmahip_conclave_records(self.h, self.ws, n, _p(ql), _p(ql2), _p(of), C.byref(hs), C.byref(oc)))
        for key in ("tmpl", "start", "end"):
            o[key] = o[key][:n]
        return o

    def conclave_se_dev(self, length, T_off, n_hits, best_score, h_tmpl, h_start, h_end, aln_scores, uniq_scores,
                        o_tmpl, o_start, o_end, w_scores, fragment_counts=None, read_counts=None, depth=None, stream=None):
        """Stage 3b on device tensors (outputs of align_se_dev); asynchronous on `stream`."""
        dp = lambda t: t.data_ptr() if t is not None else None
        r = Reads(length.numel(), None, None, length.data_ptr(), None, None, 0, 0, 0)
        c = Cands(None, None, T_off.data_ptr(), None, 0)
        h = Hits(n_hits.data_ptr(), best_score.data_ptr(), None, h_tmpl.data_ptr(), None, h_start.data_ptr(), h_end.data_ptr(),
                 aln_scores.data_ptr(), uniq_scores.data_ptr(), None)
        o = Conclave(o_tmpl.data_ptr(), o_start.data_ptr(), o_end.data_ptr(), w_scores.data_ptr(), dp(fragment_counts),
                     dp(read_counts), dp(depth))
        _check(lib().kmahip_conclave_se_dev(self.h, self.ws, C.byref(r), C.byref(c), C.byref(h), C.byref(o), C.c_void_p(stream or 0)))

    def align_trace(self, batch, rc, tmpl, tmpl_ok=None):
        """Stage 3c per read (host arrays): traceback alignment of every read against the template ConClave chose.
        -> stats [n, 10] (score, start, end, aln_len, clip_start, clip_end, match, tGaps, qGaps, mapQ; zeros = dropped),
        ops_off [n], n_ops [n], ops (runs (len << 2) | class, classes = X I D)"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        fl = np.ascontiguousarray(rc if n else np.zeros(1, np.int32), np.int32)
        tm = np.ascontiguousarray(tmpl if n else np.zeros(1, np.int32), np.int32)
        ok = None if tmpl_ok is None else np.ascontiguousarray(tmpl_ok, np.uint8)
        stats = np.zeros((max(n, 1), 10), np.int32)
        off = np.zeros(max(n, 1), np.int64)
        nops = np.zeros(max(n, 1), np.int32)
        cap = max(1024, 8 * n)
        p = Params.from_buffer_copy(self.params)
        need = C.c_int64()
        for _ in range(3):
            ops = np.zeros(cap, np.uint32)
            o = Traces(_p(stats), _p(off), _p(nops), _p(ops), cap)
            rc = lib().kmahip_align_trace(self.h, self.ws, C.byref(r), _p(fl), _p(tm), None if ok is None else _p(ok), C.byref(p),
                                          C.byref(o), C.byref(need))
            if rc == -6:
                cap = need.value + 16
                continue
            _check(rc)
            return stats[:n], off[:n], nops[:n], ops[:need.value]
        raise KmaHipError("align_trace: output capacity kept overflowing")

    def align_trace_mt1(self, batch, tmpl, one2one=0):
        """`-Mt1 tmpl`: every raw read against one template, strand by anker_rc -> (stats, ops_off, n_ops, ops) as align_trace, rc [n]"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        stats = np.zeros((max(n, 1), 10), np.int32)
        off = np.zeros(max(n, 1), np.int64)
        nops = np.zeros(max(n, 1), np.int32)
        rc_out = np.zeros(max(n, 1), np.int32)
        cap = max(1024, int(batch.length.sum()) // 4)
        p = Params.from_buffer_copy(self.params)
        need = C.c_int64()
        for _ in range(3):
            ops = np.zeros(cap, np.uint32)
            o = Traces(_p(stats), _p(off), _p(nops), _p(ops), cap)
            rc = lib().kmahip_align_trace_mt1(self.h, self.ws, C.byref(r), int(tmpl), int(one2one), C.byref(p), C.byref(o), _p(rc_out), C.byref(need))
            if rc == -6 and need.value > cap:
                cap = need.value + 16
                continue
            _check(rc)
            return (stats[:n], off[:n], nops[:n], ops[:need.value]), rc_out[:n]
        raise KmaHipError("align_trace_mt1: output capacity kept overflowing")

    def assemble(self, batch, rc, tmpl, traces, max_frag=0, bcd=1, evalue=0.05, consensus=False, frag_rank=None):
        """Stage 3c per template: pile-up of the traced reads + consensus -> dict(cover, aln_len, depth, asm_len [DB_size],
        consensus {template: str} when asked). traces = the tuple align_trace returned. frag_rank: positions of the reads among
        the filed fragments of the whole stream, for a batch gathered from several read shards (kmahip_assemble_opts.frag_rank)."""
        n = batch.n
        stats, off, nops, ops = traces
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        fl = np.ascontiguousarray(rc if n else np.zeros(1, np.int32), np.int32)
        tm = np.ascontiguousarray(tmpl if n else np.zeros(1, np.int32), np.int32)
        st = np.ascontiguousarray(stats if n else np.zeros((1, 10), np.int32), np.int32)
        of = np.ascontiguousarray(off if n else np.zeros(1, np.int64), np.int64)
        no = np.ascontiguousarray(nops if n else np.zeros(1, np.int32), np.int32)
        op = np.ascontiguousarray(ops if len(ops) else np.zeros(1, np.uint32), np.uint32)
        tr = Traces(_p(st), _p(of), _p(no), _p(op), len(op))
        D = int(self.info.DB_size)
        o = dict(cover=np.zeros(D, np.int64), aln_len=np.zeros(D, np.int64), depth=np.zeros(D, np.int64), asm_len=np.zeros(D, np.int64))
        cap = 0
        cbuf = coff = None
        if consensus:
            cap = int(2 * np.fromfile(self.prefix + ".length.b", dtype=np.int32)[1:].astype(np.int64).sum() + 4 * D + (1 << 20))
            cbuf = np.zeros(cap, np.uint8)
            coff = np.full(D, -1, np.int64)
        a = Assembly(_p(o["cover"]), _p(o["aln_len"]), _p(o["depth"]), _p(o["asm_len"]), None if cbuf is None else _p(cbuf),
                     None if coff is None else _p(coff), cap, 0)
        fr = None if frag_rank is None else np.ascontiguousarray(frag_rank if n else np.zeros(1, np.int64), np.int64)
        ao = AssembleOpts(int(max_frag), float(evalue), int(bcd), 0, 0, 0, None if fr is None else _p(fr))
        _check(lib().kmahip_assemble2(self.h, self.ws, C.byref(r), _p(fl), _p(tm), C.byref(tr), C.byref(ao), C.byref(a)))
        if consensus:
            raw = cbuf.tobytes()
            o["consensus"] = {t: raw[coff[t]:raw.index(b"\0", coff[t])].decode() for t in range(D) if coff[t] >= 0}
        return o

    def run_se(self, batch, evalue=0.05, bcd=1, max_frag=0, consensus=True, per_read=True, bc_nano=False):
        """The whole single-end run in one call (kmahip_run_se) -> dict(rows [ResRow], cover, aln_len, depth, asm_len, consensus,
        tmpl, n_hits, rc, trace_stats, ms)"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        D = int(self.info.DB_size)
        o = dict(cover=np.zeros(D, np.int64), aln_len=np.zeros(D, np.int64), depth=np.zeros(D, np.int64), asm_len=np.zeros(D, np.int64))
        cap = 0
        cbuf = coff = None
        if consensus:
            cap = int(2 * np.fromfile(self.prefix + ".length.b", dtype=np.int32)[1:].astype(np.int64).sum() + 4 * D + (1 << 20))
            cbuf = np.zeros(cap, np.uint8)
            coff = np.full(D, -1, np.int64)
        rows = (ResRow * D)()
        pr = {k: np.zeros(max(1, n) * (10 if k == "trace_stats" else 1), np.int32) for k in ("tmpl", "n_hits", "rc", "trace_stats")} if per_read else {}
        run = Run(C.cast(rows, C.c_void_p), D, 0,
                  Assembly(_p(o["cover"]), _p(o["aln_len"]), _p(o["depth"]), _p(o["asm_len"]), None if cbuf is None else _p(cbuf),
                           None if coff is None else _p(coff), cap, 0),
                  *[(_p(pr[k]) if per_read else None) for k in ("tmpl", "n_hits", "rc", "trace_stats")])
        run.caller = run.sig90 = 1 if bc_nano else 0
        p = Params.from_buffer_copy(self.params)
        _check(lib().kmahip_run_se(self.h, self.ws, C.byref(r), C.byref(p), float(evalue), int(bcd), int(max_frag), C.byref(run)))
        o["rows"] = [rows[i] for i in range(run.n_rows)]
        if consensus:
            raw = cbuf.tobytes()
            o["consensus"] = {t: raw[coff[t]:raw.index(b"\0", coff[t])].decode() for t in range(D) if coff[t] >= 0}
        for k, v in pr.items():
            o[k] = v[:n * 10].reshape(n, 10) if k == "trace_stats" else v[:n]
        o["ms"] = list(run.ms)
        return o

    def run_mt1(self, batch, tmpl, one2one=0, bc_nano=True, evalue=0.05, bcd=1, consensus=True):
        """The `-Mt1 tmpl [-bcNano]` run in one call (kmahip_run_mt1) -> dict(row, cover, aln_len, depth, asm_len, consensus, tmpl, n_hits,
        rc, trace_stats, ms)"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        D = int(self.info.DB_size)
        o = dict(cover=np.zeros(D, np.int64), aln_len=np.zeros(D, np.int64), depth=np.zeros(D, np.int64), asm_len=np.zeros(D, np.int64))
        cap = 0
        cbuf = coff = None
        if consensus:
            cap = int(2 * np.fromfile(self.prefix + ".length.b", dtype=np.int32)[1:].astype(np.int64).sum() + 4 * D + (1 << 20))
            cbuf = np.zeros(cap, np.uint8)
            coff = np.full(D, -1, np.int64)
        rows = (ResRow * 1)()
        pr = {k: np.zeros(max(1, n) * (10 if k == "trace_stats" else 1), np.int32) for k in ("tmpl", "n_hits", "rc", "trace_stats")}
        run = Run(C.cast(rows, C.c_void_p), 1, 0,
                  Assembly(_p(o["cover"]), _p(o["aln_len"]), _p(o["depth"]), _p(o["asm_len"]), None if cbuf is None else _p(cbuf),
                           None if coff is None else _p(coff), cap, 0),
                  *[_p(pr[k]) for k in ("tmpl", "n_hits", "rc", "trace_stats")])
        p = Params.from_buffer_copy(self.params)
        ao = AssembleOpts(0, float(evalue), int(bcd), 1, 1 if bc_nano else 0, 1 if bc_nano else 0)
        _check(lib().kmahip_run_mt1(self.h, self.ws, C.byref(r), int(tmpl), int(one2one), C.byref(p), C.byref(ao), C.byref(run)))
        o["row"] = rows[0]
        if consensus:
            raw = cbuf.tobytes()
            o["consensus"] = {t: raw[coff[t]:raw.index(b"\0", coff[t])].decode() for t in range(D) if coff[t] >= 0}
        for k, v in pr.items():
            o[k] = v[:n * 10].reshape(n, 10) if k == "trace_stats" else v[:n]
        o["ms"] = list(run.ms)
        return o

    def frag_write2(self, path, batch, rc, tmpl, n_hits, stats, read_names, order=1, max_frag=0, frag_rank=None):
        """kmahip_frag_write3: order 1 = stream order (`-Mt1`); frag_rank as in assemble"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        fl = np.ascontiguousarray(rc if n else np.zeros(1, np.int32), np.int32)
        tm = np.ascontiguousarray(tmpl if n else np.zeros(1, np.int32), np.int32)
        nh = np.ascontiguousarray(n_hits if n else np.zeros(1, np.int32), np.int32)
        st = np.ascontiguousarray(stats if n else np.zeros((1, 10), np.int32), np.int32)
        blob = b"".join(nm + b"\0" for nm in read_names) + b"\0"
        noff = np.zeros(n + 1, np.int64)
        if n:
            noff[1:] = np.cumsum([len(nm) + 1 for nm in read_names])
        rows = C.c_int64()
        fr = None if frag_rank is None else np.ascontiguousarray(frag_rank if n else np.zeros(1, np.int64), np.int64)
        _check(lib().kmahip_frag_write3(os.fsencode(path), self.h, C.byref(r), _p(fl), _p(tm), _p(nh), _p(st), int(max_frag), int(order),
                                        None if fr is None else _p(fr), blob, _p(noff), C.byref(rows)))
        return rows.value

    def set_pe_chain(self, on=True, minlen=16, coverT=0.1, mrs=0.5):
        """Paired runs on this workspace in the reference's default mode (no -1t1): records that lost their mate go to the chain
        finder (kmahip_ws_set_pe_chain); on=False: back to -1t1."""
        _check(lib().kmahip_ws_set_pe_chain(self.ws, C.byref(ChainParams(int(minlen), 0, float(coverT), float(mrs))) if on else None))

    def run_pe(self, batch, names, pair, evalue=0.05, bcd=1, max_frag=0, frag_path=None):
        """The paired run in one call (kmahip_run_pe) on what Ingest.next returned for two mate files -> dict(rows, cover, aln_len,
        depth, asm_len, consensus, ms)"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        blob = b"".join(nm + b"\0" for nm in names) + b"\0"
        blob_buf = C.create_string_buffer(blob, len(blob))
        noff = np.zeros(n + 1, np.int64)
        if n:
            noff[1:] = np.cumsum([len(nm) + 1 for nm in names])
        pr = np.ascontiguousarray(pair, np.uint8)
        rb = ReadBatchC(Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                              int(batch.length.max()) if n else 0), C.cast(blob_buf, C.c_void_p), _p(noff), _p(pr), 0)
        D = int(self.info.DB_size)
        o = dict(cover=np.zeros(D, np.int64), aln_len=np.zeros(D, np.int64), depth=np.zeros(D, np.int64), asm_len=np.zeros(D, np.int64))
        cap = int(2 * np.fromfile(self.prefix + ".length.b", dtype=np.int32)[1:].astype(np.int64).sum() + 4 * D + (1 << 20))
        cbuf = np.zeros(cap, np.uint8)
        coff = np.full(D, -1, np.int64)
        rows = (ResRow * D)()
        run = Run(C.cast(rows, C.c_void_p), D, 0, Assembly(_p(o["cover"]), _p(o["aln_len"]), _p(o["depth"]), _p(o["asm_len"]), _p(cbuf), _p(coff), cap, 0),
                  None, None, None, None)
        p = Params.from_buffer_copy(self.params)
        _check(lib().kmahip_run_pe(self.h, self.ws, C.byref(rb), C.byref(p), float(evalue), int(bcd), int(max_frag),
                                   os.fsencode(frag_path) if frag_path else None, C.byref(run)))
        o["rows"] = [rows[i] for i in range(run.n_rows)]
        raw = cbuf.tobytes()
        o["consensus"] = {t: raw[coff[t]:raw.index(b"\0", coff[t])].decode() for t in range(D) if coff[t] >= 0}
        o["ms"] = list(run.ms)
        return o

    def frag_write(self, path, batch, rc, tmpl, n_hits, stats, read_names, max_frag=0):
        """`.frag.gz` of the traced reads (kmahip_frag_write); read_names = list[bytes] (Ingest.next) -> number of rows"""
        n = batch.n
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        fl = np.ascontiguousarray(rc if n else np.zeros(1, np.int32), np.int32)
        tm = np.ascontiguousarray(tmpl if n else np.zeros(1, np.int32), np.int32)
        nh = np.ascontiguousarray(n_hits if n else np.zeros(1, np.int32), np.int32)
        st = np.ascontiguousarray(stats if n else np.zeros((1, 10), np.int32), np.int32)
        blob = b"".join(nm + b"\0" for nm in read_names) + b"\0"
        noff = np.zeros(n + 1, np.int64)
        if n:
            noff[1:] = np.cumsum([len(nm) + 1 for nm in read_names])
        rows = C.c_int64()
        _check(lib().kmahip_frag_write(os.fsencode(path), self.h, C.byref(r), _p(fl), _p(tm), _p(nh), _p(st), int(max_frag), blob,
                                       _p(noff), C.byref(rows)))
        return rows.value

    @staticmethod
    def res_line(name, row, cover, aln_len, depth, ID_t=1.0, Depth_t=0.0):
        """The `.res` row as the reference prints it, or None when it prints none"""
        buf = C.create_string_buffer(4096 + len(name))
        n = lib().kmahip_res_line(name.encode(), C.byref(row), int(cover), int(aln_len), int(depth), ID_t, Depth_t, buf, len(buf))
        return buf.raw[:n].decode() if n else None

    def res_rows(self, w_scores, evalue=0.05, scoreT=0.5):
        """Leading `.res` columns per template with a score -> list of ResRow (host arithmetic, runkma.c:765-783)"""
        w = np.ascontiguousarray(w_scores, np.uint64)
        cap = int((w[1:] > 0).sum()) + 1
        rows = (ResRow * cap)()
        n = C.c_int64()
        _check(lib().kmahip_res_rows(self.h, _p(w), C.c_double(evalue), C.c_double(scoreT), rows, cap, C.byref(n)))
        return [rows[i] for i in range(n.value)]

    # -- paired end stage 2 (-apm p), host buffers ------------------------------------
    def scan_pe(self, batch, exhaustive=0, t_cap=None):
        """batch = mates interleaved (read 2i, 2i+1 = pair i) -> mate, rc, rc_flag, flag [2*pairs], R_off, T"""
        n = batch.n
        assert n % 2 == 0
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        mate, rc, rc_flag, flag = (np.zeros(max(n, 1), np.int32) for _ in range(4))
        R_off = np.zeros(n + 1, np.int64)
        cap = t_cap or max(1024, 16 * n)
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        for _ in range(6):
            T = np.zeros(cap, np.int32)
            out = PeRecs(_p(mate), _p(rc), _p(rc_flag), _p(flag), _p(R_off), _p(T), cap)
            rcode = lib().kmahip_scan_pe(self.h, self.ws, C.byref(r), C.byref(p), C.byref(out))
            if rcode == -6:
                cap = max(cap * 2, int(R_off[n]) + 16)
                continue
            _check(rcode)
            return mate[:n], rc[:n], rc_flag[:n], flag[:n], R_off, T[:R_off[n]]
        raise KmaHipError("scan_pe: output capacity kept overflowing")

    def map_pe(self, batch, exhaustive=0, t_cap=None):
        """Stages 2 + 3a for interleaved mates -> (mate, rc, rc_flag, flag, R_off, T), hits dict (+ "kind" per pair)"""
        n = batch.n
        assert n % 2 == 0
        seq = np.ascontiguousarray(batch.seq, np.uint64)
        Nn = np.ascontiguousarray(batch.N if len(batch.N) else np.zeros(1, np.int32), np.int32)
        r = Reads(n, _p(seq), _p(batch.seq_off), _p(batch.length), _p(Nn), _p(batch.N_off), len(seq), len(batch.N),
                  int(batch.length.max()) if n else 0)
        mate, rc, rc_flag, flag = (np.zeros(max(n, 1), np.int32) for _ in range(4))
        R_off = np.zeros(n + 1, np.int64)
        cap = t_cap or max(1024, 16 * n)
        p = Params.from_buffer_copy(self.params)
        p.exhaustive = exhaustive
        D = int(self.info.DB_size)
        for _ in range(6):
            T = np.zeros(cap, np.int32)
            h = dict(n_hits=np.zeros(max(n, 1), np.int32), best_score=np.zeros(max(n, 1), np.int32),
                     flag=np.zeros(max(n, 1), np.int32), tmpl=np.zeros(cap, np.int32), score=np.zeros(cap, np.int32),
                     start=np.zeros(cap, np.int32), end=np.zeros(cap, np.int32), kind=np.zeros(max(n // 2, 1), np.int32),
                     alignment_scores=np.zeros(D, np.uint64), uniq_alignment_scores=np.zeros(D, np.uint64),
                     rc=np.zeros(max(n, 1), np.int32))
            out = PeRecs(_p(mate), _p(rc), _p(rc_flag), _p(flag), _p(R_off), _p(T), cap)
            hs = Hits(_p(h["n_hits"]), _p(h["best_score"]), _p(h["flag"]), _p(h["tmpl"]), _p(h["score"]), _p(h["start"]),
                      _p(h["end"]), _p(h["alignment_scores"]), _p(h["uniq_alignment_scores"]), _p(h["rc"]))
            rcode = lib().kmahip_map_pe(self.h, self.ws, C.byref(r), C.byref(p), C.byref(out), C.byref(hs), _p(h["kind"]))
            if rcode == -6:
                cap = max(cap * 2, int(R_off[n]) + 16)
                continue
            _check(rcode)
            return (mate[:n], rc[:n], rc_flag[:n], flag[:n], R_off, T[:R_off[n]]), h
        raise KmaHipError("map_pe: output capacity kept overflowing")
