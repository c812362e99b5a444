"""Test-side helpers: ctypes access to the CPU oracle (oracle/liboracle.so), to the
compiled reference (oracle/_ref/, build container only / prebuilt on the GPU box), PAF/FASTA
text writers and parsers for the reference's four output files.

Nothing here is imported by the product (raft_amd/, the raft CLI, libraft_hip.so).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from raft_amd.params import RaftParams  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "raft")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libraft_ref.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

ERR_NAMES = {0: "OK", 1: "PARAM", 2: "READ_ID", 3: "COORD", 4: "FRAGMENT", 5: "NOMEM"}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code} ({ERR_NAMES.get(code, '?')})")
        self.code = code


class _OParams(C.Structure):
    _fields_ = [("reso", C.c_int32), ("est_cov", C.c_int32), ("cov_mul", C.c_double),
                ("repeat_length", C.c_int32), ("interval_length", C.c_int32), ("read_length", C.c_int32),
                ("overlap_length", C.c_int32), ("flanking_length", C.c_int32)]


class _OResult(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("symmetric", C.c_int32), ("high_cov", C.c_int32),
                ("n_intervals", C.c_int64), ("total_coverage", C.c_int64), ("total_windows", C.c_int64),
                ("total_repeat_length", C.c_int64), ("total_read_length", C.c_int64),
                ("cov_offset", C.POINTER(C.c_int64)), ("cov", C.POINTER(C.c_int32)),
                ("rep_offset", C.POINTER(C.c_int64)), ("rep_s", C.POINTER(C.c_int32)), ("rep_e", C.POINTER(C.c_int32)),
                ("cut_offset", C.POINTER(C.c_int64)), ("cuts", C.POINTER(C.c_int32)),
                ("frag_offset", C.POINTER(C.c_int64)), ("frag_read", C.POINTER(C.c_int32)),
                ("frag_begin", C.POINTER(C.c_int32)), ("frag_end", C.POINTER(C.c_int32))]


def build_oracle():
    """(Re)builds oracle/liboracle.so and, when /root/reference is present, oracle/_ref/*."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_olib = None


def oracle_lib():
    global _olib
    if _olib is None:
        src = os.path.join(ORACLE_DIR, "raft_oracle.c")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            build_oracle()
        _olib = C.CDLL(ORACLE_SO)
        _olib.raft_oracle_run.argtypes = [C.POINTER(_OParams), C.c_int32, C.c_void_p, C.c_int64] + [C.c_void_p] * 6 + [C.POINTER(_OResult)]
        _olib.raft_oracle_free.argtypes = [C.POINTER(_OResult)]
        _olib.raft_oracle_free.restype = None
    return _olib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _np(ptr, n, dt):
    if n == 0:
        return np.empty(0, dt)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)


def oracle_run(p: RaftParams, read_len, qid, qs, qe, tid, ts, te) -> dict:
    """Runs the C restatement; returns numpy CSR arrays + scalars, raises OracleError."""
    lib = oracle_lib()
    cols = [_i32(a) for a in (read_len, qid, qs, qe, tid, ts, te)]
    op = _OParams(p.reso, p.est_cov, p.cov_mul, p.repeat_length, p.interval_length, p.read_length,
                  p.overlap_length, p.flanking_length)
    res = _OResult()
    rc = lib.raft_oracle_run(C.byref(op), cols[0].size, cols[0].ctypes.data, cols[1].size,
                             *[a.ctypes.data for a in cols[1:]], C.byref(res))
    if rc != 0:
        raise OracleError(rc)
    n = res.n_reads
    out = {k: int(getattr(res, k)) for k in ("n_reads", "symmetric", "high_cov", "n_intervals", "total_coverage",
                                              "total_windows", "total_repeat_length", "total_read_length")}
    out["cov_offset"] = _np(res.cov_offset, n + 1, np.int64)
    out["cov"] = _np(res.cov, int(out["cov_offset"][-1]), np.int32)
    out["rep_offset"] = _np(res.rep_offset, n + 1, np.int64)
    nr = int(out["rep_offset"][-1])
    out["rep_s"], out["rep_e"] = _np(res.rep_s, nr, np.int32), _np(res.rep_e, nr, np.int32)
    out["cut_offset"] = _np(res.cut_offset, n + 1, np.int64)
    out["cuts"] = _np(res.cuts, int(out["cut_offset"][-1]), np.int32)
    out["frag_offset"] = _np(res.frag_offset, n + 1, np.int64)
    nf = int(out["frag_offset"][-1])
    out["frag_read"], out["frag_begin"], out["frag_end"] = (_np(res.frag_read, nf, np.int32), _np(res.frag_begin, nf, np.int32),
                                                           _np(res.frag_end, nf, np.int32))
    lib.raft_oracle_free(C.byref(res))
    return out


ARRAY_KEYS = ("cov_offset", "cov", "rep_offset", "rep_s", "rep_e", "cut_offset", "cuts", "frag_offset", "frag_read",
              "frag_begin", "frag_end")
SCALAR_KEYS = ("symmetric", "high_cov", "total_coverage", "total_windows", "total_repeat_length", "total_read_length")


def assert_same_result(got: dict, want: dict, what: str = ""):
    """Bit-exact comparison of two result dicts (integer path: no tolerance)."""
    for k in SCALAR_KEYS:
        assert int(got[k]) == int(want[k]), f"{what}: scalar {k}: got {got[k]} want {want[k]}"
    for k in ARRAY_KEYS:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        assert g.shape == w.shape, f"{what}: {k} shape {g.shape} != {w.shape}"
        if not np.array_equal(g, w):
            bad = np.flatnonzero(g != w)
            raise AssertionError(f"{what}: {k} differs at {bad.size} of {g.size} entries; first at {bad[0]}: got {g[bad[0]]} want {w[bad[0]]}")


# ---- the compiled reference (build container, or prebuilt oracle/_ref on the GPU box) ---------

class _RParams(C.Structure):
    _fields_ = _OParams._fields_


def have_ref_lib():
    return os.path.exists(REF_SO)


def have_ref_bin():
    return os.path.exists(REF_BIN)


_rlib = None


def ref_lib_run(p: RaftParams, read_len, qid, qs, qe, tid, ts, te, want_cov=True) -> dict:
    """profileCoverage / repeat_annotate of the unmodified reference through oracle/ref_harness.cpp."""
    global _rlib
    if _rlib is None:
        _rlib = C.CDLL(REF_SO)
        _rlib.raft_ref_run.argtypes = [C.POINTER(_RParams), C.c_int32, C.c_void_p, C.c_int64] + [C.c_void_p] * 6 + \
            [C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
             C.POINTER(C.c_double)]
    cols = [_i32(a) for a in (read_len, qid, qs, qe, tid, ts, te)]
    rp = _RParams(p.reso, p.est_cov, p.cov_mul, p.repeat_length, p.interval_length, p.read_length,
                  p.overlap_length, p.flanking_length)
    n = cols[0].size
    nb = (cols[0].astype(np.int64) + p.reso - 1) // p.reso
    cov = np.zeros(int(nb.sum()) if want_cov else 0, np.int32)
    rep_count = np.zeros(n, np.int32)
    cap = int(nb.sum()) + n + 1
    rep_s, rep_e = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    sym, n_rep = C.c_int32(), C.c_int64()
    secs = (C.c_double * 2)()
    rc = _rlib.raft_ref_run(C.byref(rp), n, cols[0].ctypes.data, cols[1].size, *[a.ctypes.data for a in cols[1:]],
                            C.byref(sym), cov.ctypes.data if want_cov else None, rep_count.ctypes.data,
                            rep_s.ctypes.data, rep_e.ctypes.data, cap, C.byref(n_rep), secs)
    assert rc == 0
    rep_offset = np.zeros(n + 1, np.int64)
    np.cumsum(rep_count, out=rep_offset[1:])
    return {"symmetric": sym.value, "cov": cov, "rep_offset": rep_offset, "rep_s": rep_s[:n_rep.value].copy(),
            "rep_e": rep_e[:n_rep.value].copy(), "seconds_bucket": secs[0], "seconds_annotate": secs[1]}


# ---- text I/O in the reference's formats ----------------------------------------------------------

def seq_of(n: int, salt: int = 0) -> str:
    """Deterministic bases of length n (content never influences the path; only FASTA slices echo it)."""
    unit = "ACGTTGCAAGCTGATC"
    k = salt % len(unit)
    unit = unit[k:] + unit[:k]
    return (unit * (n // len(unit) + 1))[:n]


def write_fasta(path, names, lens):
    with open(path, "w") as f:
        for i, (nm, ln) in enumerate(zip(names, lens)):
            f.write(f">{nm}\n{seq_of(int(ln), i)}\n")


def write_paf(path, names, read_len, qid, qs, qe, tid, ts, te):
    rl = np.asarray(read_len)
    with open(path, "w") as f:
        for a, b, c, d, e, g in zip(qid, qs, qe, tid, ts, te):
            f.write(f"{names[a]}\t{rl[a]}\t{b}\t{c}\t+\t{names[d]}\t{rl[d]}\t{e}\t{g}\t{max(c - b, 0)}\t{max(c - b, 1)}\t60\n")


def run_ref_binary(workdir, args, fasta, paf, binary=REF_BIN):
    """Runs ``raft [args] fasta paf`` in workdir; returns (returncode, stdout bytes)."""
    r = subprocess.run([binary] + list(args) + [fasta, paf], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return r.returncode, r.stdout


def parse_coverage_txt(text: str):
    """-> list of int arrays (counts per window), checking the 'pos' column is j*reso-consistent."""
    rows = []
    for i, line in enumerate(text.split("\n")[:-1]):
        toks = line.split(" ")
        assert toks[0] == "read" and int(toks[1]) == i and toks[-1] == "", line[:80]
        rows.append(np.array([int(t.split(",")[1]) for t in toks[2:-1]], np.int32))
    return rows


def parse_long_repeats(text: str):
    rows = []
    for i, line in enumerate(text.split("\n")[:-1]):
        m = re.match(r"read (\d+), (.*)$", line)
        assert m and int(m.group(1)) == i, line[:80]
        pairs = [t for t in m.group(2).split("    ") if t]
        rows.append([(int(a), int(b)) for a, b in (t.split(",") for t in pairs)])
    return rows


def parse_fasta_headers(text: str):
    """Real-read mode headers -> list of (read_num, name, begin, end)."""
    out = []
    for line in text.split("\n"):
        if line.startswith(">"):
            m = re.match(r">read=(\d+),(.*),pos_on_original_read=(-?\d+)-(-?\d+)$", line)
            assert m, line
            out.append((int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))))
    return out


def md5(b: bytes) -> str:
    return hashlib.md5(b).hexdigest()


def result_from_ref_files(prefix_path: str, names) -> dict:
    """Parses coverage.txt / long_repeats.txt / reads.fasta of a reference run into the CSR dict layout."""
    cov_rows = parse_coverage_txt(open(prefix_path + ".coverage.txt").read())
    rep_rows = parse_long_repeats(open(prefix_path + ".long_repeats.txt").read())
    hdr = parse_fasta_headers(open(prefix_path + ".reads.fasta").read())
    n = len(cov_rows)
    idx = {nm: i for i, nm in enumerate(names)}
    out = {"cov_offset": np.zeros(n + 1, np.int64), "rep_offset": np.zeros(n + 1, np.int64)}
    np.cumsum([len(r) for r in cov_rows], out=out["cov_offset"][1:])
    np.cumsum([len(r) for r in rep_rows], out=out["rep_offset"][1:])
    out["cov"] = np.concatenate(cov_rows).astype(np.int32) if n else np.empty(0, np.int32)
    flat = [pr for r in rep_rows for pr in r]
    out["rep_s"] = np.array([a for a, _ in flat], np.int32)
    out["rep_e"] = np.array([b for _, b in flat], np.int32)
    assert [h[0] for h in hdr] == list(range(1, len(hdr) + 1))
    out["frag_read"] = np.array([idx[h[1]] for h in hdr], np.int32)
    out["frag_begin"] = np.array([h[2] for h in hdr], np.int32)
    out["frag_end"] = np.array([h[3] for h in hdr], np.int32)
    fo = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(out["frag_read"], minlength=n), out=fo[1:])
    out["frag_offset"] = fo
    return out


# ---- shared input generators ------------------------------------------------------------------------

def tie_case(seed):
    """One or more reads with MANY short high-coverage runs near the read's begin and a flank larger than their
    distance from it: several repeats clamp to start 0 and, with more than 16 repeats in the read, libstdc++'s
    std::sort (repeat.hpp:170) permutes them."""
    rng = np.random.default_rng(1000 + seed)
    reso = int(rng.choice([1, 5, 10]))
    n_reads = int(rng.integers(1, 4))
    run_w = int(rng.integers(2, 6))                     # windows per run
    gap_w = int(rng.integers(1, 5))
    rl, rec = [], []
    for r in range(n_reads):
        n_runs = int(rng.integers(2, 120))
        pos = int(rng.integers(0, 4)) * reso
        for _ in range(n_runs):
            w = run_w + int(rng.integers(0, 3))
            rec.append((r, pos, pos + w * reso))
            pos += (w + gap_w + int(rng.integers(0, 3))) * reso
        rl.append(pos + int(rng.integers(0, 50 * reso)))
    rl = np.array(rl, np.int32)
    qid = np.array([x[0] for x in rec], np.int32)
    qs = np.array([x[1] for x in rec], np.int32)
    qe = np.array([x[2] for x in rec], np.int32)
    tid = qid.copy()                                    # self overlaps: query side only (chop.hpp:166)
    L = int(rng.choice([100, 1000]))
    p = RaftParams(reso=reso, est_cov=1, cov_mul=1.0, repeat_length=run_w * reso, interval_length=L,
                   read_length=2 * L, overlap_length=0,
                   flanking_length=int(rng.choice([0, 50, 400, 2000, 100000])))
    p.repeat_length = run_w * reso
    return p, [rl, qid, qs, qe, tid, qs.copy(), qe.copy()]


# ---- tests/golden/ref_fuzz.npz: random text inputs with the outputs of the compiled reference binary ------------

_fuzz = None


def ref_fuzz_count() -> int:
    global _fuzz
    if _fuzz is None:
        with np.load(os.path.join(GOLDEN, "ref_fuzz.npz")) as z:
            _fuzz = {k: z[k] for k in z.files}          # (an NpzFile would inflate the member again on every access)
    return int(_fuzz["seeds"].size)


def ref_fuzz_case(i: int):
    """-> (RaftParams, [read_len, qid, qs, qe, tid, ts, te], expected dict incl. 'symmetric', 'stats', 'md5')."""
    ref_fuzz_count()
    z = _fuzz
    reso, est_cov, rep_len, iv_len, read_length, ovl, flank = (int(x) for x in z["params"][i])
    p = RaftParams(reso=reso, est_cov=est_cov, cov_mul=float(z["cov_mul"][i]), repeat_length=rep_len, interval_length=iv_len,
                   read_length=read_length, overlap_length=ovl, flanking_length=flank)
    r0, r1 = (int(x) for x in z["off_reads"][i:i + 2])
    c0, c1 = (int(x) for x in z["off_recs"][i:i + 2])
    cols = [z["read_len"][r0:r1]] + [z[k][c0:c1] for k in ("qid", "qs", "qe", "tid", "ts", "te")]
    b0, b1 = (int(x) for x in z["off_cov"][i:i + 2])
    p0, p1 = (int(x) for x in z["off_rep"][i:i + 2])
    f0, f1 = (int(x) for x in z["off_frag"][i:i + 2])
    rep_offset = np.zeros(r1 - r0 + 1, np.int64)
    np.cumsum(z["rep_cnt"][r0:r1], out=rep_offset[1:])
    exp = {"cov": z["cov"][b0:b1], "rep_offset": rep_offset, "rep_s": z["rep_s"][p0:p1], "rep_e": z["rep_e"][p0:p1],
           "frag_read": z["frag_read"][f0:f1], "frag_begin": z["frag_begin"][f0:f1], "frag_end": z["frag_end"][f0:f1],
           "symmetric": int(z["symmetric"][i]), "stats": str(z["stats"][i]),
           "md5": dict(zip(("reads.fasta", "coverage.txt", "long_repeats.txt", "long_repeats.bed"), (str(x) for x in z["md5"][i])))}
    return p, cols, exp


def assert_matches_ref_fuzz(got: dict, exp: dict, p: RaftParams, what: str = ""):
    """Result dict (oracle or engine) against the parsed outputs + stdout statistics of the reference binary."""
    assert int(got["symmetric"]) == exp["symmetric"], what
    for k in ("cov", "rep_offset", "rep_s", "rep_e", "frag_read", "frag_begin", "frag_end"):
        assert np.array_equal(np.asarray(got[k]), exp[k]), (what, k)
    if int(got["total_windows"]) > 0 and int(got["total_read_length"]) > 0:   # the reference prints nan / inf otherwise
        cpw = got["total_coverage"] / got["total_windows"]
        want = ("coverage per window is %f \n" % cpw + "coverage per window/average coverage is %f \n" % (cpw / p.est_cov) +
                "fraction_of_repeat_length %f " % (got["total_repeat_length"] / got["total_read_length"]))
        assert want == exp["stats"], (what, want, exp["stats"])


# ---- BASELINE configs[0] stand-in (tests/golden/c1_chr11_standin.npz, make_golden.py config1_case) ---------------

def load_config1():
    import json
    meta = json.load(open(os.path.join(GOLDEN, "manifest.json")))["config1"]["c1_chr11_standin"]
    with np.load(os.path.join(GOLDEN, "c1_chr11_standin.npz")) as z:
        cols = [z[k] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te")]
        exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    return RaftParams(**meta["params"]), cols, exp, meta


def write_config1_inputs(tmp, cols, meta):
    """reads.fa.gz + overlaps.paf.gz exactly as make_golden.py wrote them for the reference run (the md5 of
    fragmented.reads.fasta depends on the bases, which seq_of() regenerates)."""
    import gzip
    import shutil
    names = [meta["name_format"].format(i=i) for i in range(len(cols[0]))]
    write_fasta(os.path.join(tmp, "reads.fa"), names, cols[0])
    write_paf(os.path.join(tmp, "overlaps.paf"), names, *cols)
    for f in ("reads.fa", "overlaps.paf"):
        with open(os.path.join(tmp, f), "rb") as i, gzip.open(os.path.join(tmp, f + ".gz"), "wb", compresslevel=1) as z:
            shutil.copyfileobj(i, z)
        os.remove(os.path.join(tmp, f))
    return names


# ---- the two pileup kernels ---------------------------------------------------------------------------------------------------
# Rounds 1-5 kept several pileup kernels and the suites ran every case through each of them (raft_hip_set_tuning's `variant`).  Since
# round 6 there is the wave kernel (16-bit difference array, one wave per tile: raft_amd/csrc/pileup_wave.hpp) and, for tiles of 2^15
# intervals or more, the 32-bit side kernel (pileup_deep.hpp).  KERNELS names the two ways a case can be run: as it comes, and with
# every tile sent the deep kernel's way (RAFT_DEEP_MIN=1, read by the engine at every pass) -- two independent implementations of
# repeat.hpp:28-79 / 111-168 that must agree with the oracle and with each other.
KERNELS = ("wave", "deep")


class kernel_mode:
    """with kernel_mode("deep"): every pass started inside sends all its tiles through pileup_deep_kernel."""
    def __init__(self, mode):
        assert mode in KERNELS or mode in (-1, 5, None), mode
        self.deep = mode == "deep"

    def __enter__(self):
        self.old = os.environ.get("RAFT_DEEP_MIN")
        if self.deep:
            os.environ["RAFT_DEEP_MIN"] = "1"
        return self

    def __exit__(self, *exc):
        if self.deep:
            if self.old is None:
                os.environ.pop("RAFT_DEEP_MIN", None)
            else:
                os.environ["RAFT_DEEP_MIN"] = self.old
        return False
