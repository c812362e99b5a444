"""ctypes binding of include/raft_host.h (libraft_host.so): the host text layer of the `raft` CLI."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB_PATH = os.environ.get("RAFT_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libraft_host.so")   # (override: sanitizer builds)
OK, ERR_OPEN, ERR_DUP_NAME, ERR_UNKNOWN_NAME, ERR_IO, ERR_ARG = range(6)

EXPORTS = ("raft_host_reads_load", "raft_host_reads_free", "raft_host_reads_count", "raft_host_reads_lengths",
           "raft_host_reads_name", "raft_host_reads_bases", "raft_host_reads_real", "raft_host_paf_load", "raft_host_paf_free",
           "raft_host_paf_count", "raft_host_paf_column", "raft_host_write_coverage", "raft_host_write_repeats",
           "raft_host_write_fasta", "raft_host_set_threads", "raft_host_get_threads", "raft_host_split_naive",
           "raft_host_paf_symmetric", "raft_host_unpack_coverage", "raft_host_write_coverage_packed",
           "raft_host_unpack_coverage_w", "raft_host_write_coverage_packed_w", "raft_host_text_read", "raft_host_text_free",
           "raft_host_paf_parse", "raft_host_group_offsets", "raft_host_pack_windows", "raft_host_unpack_coverage_d4",
           "raft_host_write_coverage_d4")


class HostError(RuntimeError):
    def __init__(self, code, what=""):
        super().__init__(f"raft_host error {code} {what}")
        self.code = code
        self.what = what


_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        lib.raft_host_reads_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        lib.raft_host_reads_free.argtypes = [vp]; lib.raft_host_reads_free.restype = None
        lib.raft_host_reads_count.argtypes = [vp]; lib.raft_host_reads_count.restype = C.c_int32
        lib.raft_host_reads_lengths.argtypes = [vp]; lib.raft_host_reads_lengths.restype = C.POINTER(C.c_int32)
        lib.raft_host_reads_name.argtypes = [vp, C.c_int32]; lib.raft_host_reads_name.restype = C.c_char_p
        lib.raft_host_reads_bases.argtypes = [vp, C.c_int32]; lib.raft_host_reads_bases.restype = C.POINTER(C.c_char)
        lib.raft_host_reads_real.argtypes = [vp]
        lib.raft_host_paf_load.argtypes = [C.c_char_p, vp, C.POINTER(vp), C.c_char_p, C.c_int]
        lib.raft_host_text_read.argtypes = [C.c_char_p, C.POINTER(vp)]
        lib.raft_host_text_free.argtypes = [vp]
        lib.raft_host_text_free.restype = None
        lib.raft_host_paf_parse.argtypes = [vp, vp, C.POINTER(vp), C.c_char_p, C.c_int]
        lib.raft_host_paf_free.argtypes = [vp]; lib.raft_host_paf_free.restype = None
        lib.raft_host_paf_count.argtypes = [vp]; lib.raft_host_paf_count.restype = C.c_int64
        lib.raft_host_paf_column.argtypes = [vp, C.c_int]; lib.raft_host_paf_column.restype = C.POINTER(C.c_int32)
        lib.raft_host_write_coverage.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp, vp]
        lib.raft_host_write_repeats.argtypes = [C.c_char_p, C.c_char_p, vp, vp, vp, vp]
        lib.raft_host_write_fasta.argtypes = [C.c_char_p, vp, vp, vp, vp]
        lib.raft_host_split_naive.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]
        lib.raft_host_set_threads.argtypes = [C.c_int]
        lib.raft_host_paf_symmetric.argtypes = [vp]
        lib.raft_host_unpack_coverage.argtypes = [C.c_int64, vp, C.c_int64, vp, vp, vp]
        lib.raft_host_write_coverage_packed.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp]
        lib.raft_host_unpack_coverage_w.argtypes = [C.c_int32, C.c_int64, vp, C.c_int64, vp, vp, vp]
        lib.raft_host_write_coverage_packed_w.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_int32, vp, vp, C.c_int64, vp, vp]
        lib.raft_host_get_threads.argtypes = []
        lib.raft_host_group_offsets.argtypes = [C.c_int32, C.c_int64, vp, C.c_int32, C.POINTER(C.c_int32), vp]
        lib.raft_host_unpack_coverage_d4.argtypes = [C.c_int64, vp, vp, C.c_int64, vp, vp, vp]
        lib.raft_host_write_coverage_d4.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp, vp, vp, C.c_int64, vp, vp]
        lib.raft_host_pack_windows.argtypes = [C.c_int64, vp, vp, C.c_int32, vp, C.POINTER(C.c_int64)]
        _lib = lib
    return _lib


def set_threads(n: int) -> None:
    """Worker threads of the loaders/writers (0 = default, 1 = sequential); outputs do not depend on it."""
    rc = load_library().raft_host_set_threads(int(n))
    if rc != OK:
        raise HostError(rc, f"set_threads({n})")


def group_offsets(n_reads: int, qid, max_runs: int = 4, out=None):
    """raft_host_group_offsets: per-read record offsets of a query column that is at most ``max_runs`` runs sorted by read
    id -> int64 array [n_runs, n_reads + 1]; None when the column is not of that shape (more runs, ids out of range).
    ``out``: a caller-owned int64 array of at least max_runs * (n_reads + 1) entries (e.g. page-locked) to fill."""
    q = np.ascontiguousarray(np.asarray(qid), dtype=np.int32)
    need = max_runs * (n_reads + 1)
    buf = out if out is not None else np.empty(need, np.int64)
    if buf.dtype != np.int64 or buf.size < need or not buf.flags["C_CONTIGUOUS"]:
        raise ValueError("group_offsets: out must be a contiguous int64 array of max_runs * (n_reads + 1) entries")
    n_runs = C.c_int32(0)
    rc = load_library().raft_host_group_offsets(int(n_reads), int(q.size), C.c_void_p(q.ctypes.data if q.size else 0), int(max_runs),
                                                C.byref(n_runs), C.c_void_p(buf.ctypes.data))
    if rc != OK:
        raise HostError(rc, "group_offsets")
    if n_runs.value == 0:
        return None
    return buf[: n_runs.value * (n_reads + 1)].reshape(n_runs.value, n_reads + 1)


ERR_COORD, ERR_RANGE = 6, 7


def pack_windows(qs, qe, reso: int, out=None):
    """raft_host_pack_windows: window records (uint32: first window | one past the last << 16) of the coordinate columns, for
    the engine's ``*_windows`` entries.  Returns the array, or None when some interval ends beyond window 65,535 (the caller
    keeps the coordinate columns).  A negative coordinate raises HostError(ERR_COORD) whose ``index`` names the record.
    ``out``: a caller-owned uint32 array of at least len(qs) entries (e.g. page-locked) to fill."""
    a = np.ascontiguousarray(np.asarray(qs), dtype=np.int32)
    b = np.ascontiguousarray(np.asarray(qe), dtype=np.int32)
    if a.size != b.size:
        raise ValueError("pack_windows: qs / qe differ in length")
    buf = out if out is not None else np.empty(a.size, np.uint32)
    if buf.dtype != np.uint32 or buf.size < a.size or not buf.flags["C_CONTIGUOUS"]:
        raise ValueError("pack_windows: out must be a contiguous uint32 array of len(qs) entries")
    bad = C.c_int64(-1)
    P = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
    rc = load_library().raft_host_pack_windows(int(a.size), P(a), P(b), int(reso), P(buf), C.byref(bad))
    if rc == ERR_RANGE:
        return None
    if rc != OK:
        e = HostError(rc, f"pack_windows (record {bad.value})")
        e.index = int(bad.value)
        raise e
    return buf[: a.size]


def split_naive(in_path: str, out_path: str, split_len: int) -> int:
    """split_naive.cpp: fixed-length pieces of every read; returns the number of input records."""
    n = C.c_int32(0)
    rc = load_library().raft_host_split_naive(in_path.encode(), out_path.encode(), int(split_len), C.byref(n))
    if rc != OK:
        raise HostError(rc, in_path)
    return int(n.value)


def get_threads() -> int:
    return int(load_library().raft_host_get_threads())


class Reads:
    def __init__(self, path: str):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.raft_host_reads_load(path.encode(), C.byref(self._h))
        if rc != OK:
            raise HostError(rc, path)
        n = self._lib.raft_host_reads_count(self._h)
        self.n = n
        self.lengths = np.ctypeslib.as_array(self._lib.raft_host_reads_lengths(self._h), shape=(n,)).copy() if n else np.empty(0, np.int32)
        self.real = int(self._lib.raft_host_reads_real(self._h))

    def name(self, i: int) -> str:
        return self._lib.raft_host_reads_name(self._h, i).decode()

    def bases(self, i: int) -> bytes:
        return C.string_at(self._lib.raft_host_reads_bases(self._h, i), int(self.lengths[i]))

    def close(self):
        if self._h:
            self._lib.raft_host_reads_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_paf(path: str, reads: Reads, with_flag: bool = False, two_steps: bool = False):
    """-> six int32 numpy columns (qid, qs, qe, tid, ts, te) [, the symmetric flag the tokeniser found].
    ``two_steps``: raft_host_text_read + raft_host_paf_parse (what the CLI does, the first beside the loading of the reads)."""
    lib = load_library()
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    if two_steps:
        t = C.c_void_p()
        rc = lib.raft_host_text_read(path.encode(), C.byref(t))
        if rc == OK:
            rc = lib.raft_host_paf_parse(t, reads._h, C.byref(h), err, 256)
        lib.raft_host_text_free(t)
    else:
        rc = lib.raft_host_paf_load(path.encode(), reads._h, C.byref(h), err, 256)
    if rc != OK:
        raise HostError(rc, err.value.decode())
    n = lib.raft_host_paf_count(h)
    cols = [np.ctypeslib.as_array(lib.raft_host_paf_column(h, k), shape=(n,)).copy() if n else np.empty(0, np.int32) for k in range(6)]
    sym = int(lib.raft_host_paf_symmetric(h))
    lib.raft_host_paf_free(h)
    return (cols, sym) if with_flag else cols


def pack_coverage(cov, width: int = 1):
    """Reference encoder of the transfer form (tests): code = min(cov, limit) + ascending exceptions (index, value);
    width 1: uint8, limit 255; width 2: uint16, limit 65535."""
    cov = np.asarray(cov, np.int32)
    limit, dt = (255, np.uint8) if width == 1 else (65535, np.uint16)
    idx = np.flatnonzero(cov >= limit).astype(np.int64)
    return np.minimum(cov, limit).astype(dt), idx, cov[idx].astype(np.int32)


def _width_of(code) -> int:
    return 2 if np.asarray(code).dtype == np.uint16 else 1


def unpack_coverage(code, exc_index, exc_value):
    """raft_host_unpack_coverage_w: the int32 coverage array from the packed form (uint8 or uint16 codes)."""
    lib = load_library()
    width = _width_of(code)
    code = np.ascontiguousarray(code, np.uint16 if width == 2 else np.uint8)
    xi, xv = np.ascontiguousarray(exc_index, np.int64), np.ascontiguousarray(exc_value, np.int32)
    out = np.empty(code.size, np.int32)
    rc = lib.raft_host_unpack_coverage_w(width, code.size, C.c_void_p(code.ctypes.data), xi.size, C.c_void_p(xi.ctypes.data),
                                         C.c_void_p(xv.ctypes.data), C.c_void_p(out.ctypes.data))
    if rc != OK:
        raise HostError(rc, "unpack_coverage")
    return out


def unpack_coverage_d4(n_bins: int, cov_nib, cov_anchor, exc_index, exc_value):
    """raft_host_unpack_coverage_d4: the int32 coverage array from the four-bit step encoding."""
    nib = np.ascontiguousarray(cov_nib, np.uint8)
    an = np.ascontiguousarray(cov_anchor, np.int32)
    xi, xv = np.ascontiguousarray(exc_index, np.int64), np.ascontiguousarray(exc_value, np.int32)
    if nib.size < (n_bins + 1) // 2 or an.size < (n_bins + 1023) // 1024:
        raise ValueError("unpack_coverage_d4: cov_nib / cov_anchor too short")
    out = np.empty(n_bins, np.int32)
    p = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
    rc = load_library().raft_host_unpack_coverage_d4(int(n_bins), p(nib), p(an), xi.size, p(xi), p(xv), p(out))
    if rc != OK:
        raise HostError(rc, "unpack_coverage_d4")
    return out


def write_coverage_d4(path: str, n_reads: int, reso: int, cov_offset, cov_nib, cov_anchor, exc_index, exc_value):
    co = np.ascontiguousarray(cov_offset, np.int64)
    nib, an = np.ascontiguousarray(cov_nib, np.uint8), np.ascontiguousarray(cov_anchor, np.int32)
    xi, xv = np.ascontiguousarray(exc_index, np.int64), np.ascontiguousarray(exc_value, np.int32)
    p = lambda x: C.c_void_p(x.ctypes.data if x.size else 0)
    rc = load_library().raft_host_write_coverage_d4(path.encode(), n_reads, reso, p(co), p(nib), p(an), xi.size, p(xi), p(xv))
    if rc != OK:
        raise HostError(rc, path)


def write_coverage_packed(path: str, n_reads: int, reso: int, cov_offset, code, exc_index, exc_value):
    lib = load_library()
    width = _width_of(code)
    co = np.ascontiguousarray(cov_offset, np.int64)
    code = np.ascontiguousarray(code, np.uint16 if width == 2 else np.uint8)
    xi, xv = np.ascontiguousarray(exc_index, np.int64), np.ascontiguousarray(exc_value, np.int32)
    p = lambda x: C.c_void_p(x.ctypes.data)
    rc = lib.raft_host_write_coverage_packed_w(width, path.encode(), n_reads, reso, p(co), p(code), xi.size, p(xi), p(xv))
    if rc != OK:
        raise HostError(rc, path)


def write_outputs(prefix: str, reads: Reads, reso: int, res: dict):
    """Writes PREFIX.coverage.txt / .long_repeats.txt / .long_repeats.bed / .reads.fasta from CSR arrays."""
    lib = load_library()
    a = {k: np.ascontiguousarray(res[k]) for k in ("cov_offset", "cov", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end")}
    p = lambda x: C.c_void_p(x.ctypes.data)
    for rc in (lib.raft_host_write_coverage((prefix + ".coverage.txt").encode(), reads.n, reso, p(a["cov_offset"]), p(a["cov"])),
               lib.raft_host_write_repeats((prefix + ".long_repeats.txt").encode(), (prefix + ".long_repeats.bed").encode(), reads._h,
                                           p(a["rep_offset"]), p(a["rep_s"]), p(a["rep_e"])),
               lib.raft_host_write_fasta((prefix + ".reads.fasta").encode(), reads._h, p(a["frag_offset"]), p(a["frag_begin"]), p(a["frag_end"]))):
        if rc != OK:
            raise HostError(rc, prefix)
