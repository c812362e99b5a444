"""CPU: the C restatement (oracle/) against the golden outputs of the compiled reference.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py running
oracle/_ref/raft (the unmodified reference, Makefile:2-6) in the build container.
"""
import json
import os

import numpy as np
import pytest
from raft_testlib import (GOLDEN, OracleError, RaftParams, assert_same_result, have_ref_lib, oracle_run,
                          parse_coverage_txt, parse_fasta_headers, parse_long_repeats, ref_lib_run, tie_case)

MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))
SYNTH = sorted(MAN["synthetic"])


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = MAN["synthetic"][name]
    p = RaftParams(**meta["params"])
    cols = [z[k] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te")]
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    return p, cols, exp, meta


EXP_KEYS = ("cov_offset", "cov", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_read", "frag_begin", "frag_end")


@pytest.mark.parametrize("name", SYNTH)
def test_oracle_matches_reference_outputs(name):
    p, cols, exp, meta = load_case(name)
    got = oracle_run(p, *cols)
    assert got["symmetric"] == meta["symmetric"]
    for k in EXP_KEYS:
        assert np.array_equal(got[k], exp[k]), (name, k)
    # the reference's stdout statistics (repeat.hpp:173-178) from the oracle's totals
    out = meta["stdout"]
    cpw = got["total_coverage"] / got["total_windows"]
    assert "coverage per window is %f \n" % cpw in out
    assert "coverage per window/average coverage is %f \n" % (cpw / p.est_cov) in out
    assert "fraction_of_repeat_length %f \n" % (got["total_repeat_length"] / got["total_read_length"]) in out
    assert "high_cov %d\n" % got["high_cov"] in out
    assert "INFO, length of alignments  %d()\n" % len(cols[1]) in out


def _micro_columns(d):
    """Tokenises a micro case with python (independent of the product's C++ readers)."""
    names, lens, cur = [], [], None
    for line in open(os.path.join(d, "reads.fa"), newline="").read().replace("\r", "").split("\n"):
        if line.startswith(">"):
            names.append(line[1:].split()[0]); lens.append(0)
        elif names:
            lens[-1] += len(line.strip())
    idx = {n: i for i, n in enumerate(names)}
    rec = []
    for line in open(os.path.join(d, "overlaps.paf"), newline="").read().split("\n"):
        f = line.rstrip("\r").split("\t")
        if len(f) < 10:
            continue
        rec.append((idx[f[0]], int(f[2]), int(f[3]), idx[f[5]], int(f[7]), int(f[8])))
    cols = [np.array([r[k] for r in rec], np.int32) for k in range(6)]
    return names, np.array(lens, np.int32), cols


def _params_from_args(args):
    p = RaftParams()
    it = iter(args)
    for a in it:
        v = next(it)
        if a == "-r": p.reso = int(v)
        elif a == "-e": p.est_cov = int(v)
        elif a == "-m": p.cov_mul = float(v)
        elif a == "-l": p.read_length = int(v)
        elif a == "-p": p.repeat_length = p.interval_length = int(v)
        elif a == "-f": p.flanking_length = int(v)
        elif a == "-v": p.overlap_length = int(v)
    return p


@pytest.mark.parametrize("name", ["g1", "g2", "g4", "g5_vprefix"])
def test_oracle_micro_vectors(name):
    d = os.path.join(GOLDEN, "micro", name)
    meta = MAN["micro"][name]
    names, lens, cols = _micro_columns(d)
    p = _params_from_args(meta["args"])
    got = oracle_run(p, lens, *cols)
    prefix = [f for f in meta["outputs"] if f.endswith(".coverage.txt")][0][: -len(".coverage.txt")]
    cov_rows = parse_coverage_txt(open(os.path.join(d, f"expect.{prefix}.coverage.txt")).read())
    assert np.array_equal(got["cov"], np.concatenate(cov_rows))
    rep_rows = parse_long_repeats(open(os.path.join(d, f"expect.{prefix}.long_repeats.txt")).read())
    flat = [x for r in rep_rows for x in r]
    assert [tuple(x) for x in zip(got["rep_s"].tolist(), got["rep_e"].tolist())] == flat
    assert np.array_equal(np.diff(got["rep_offset"]), [len(r) for r in rep_rows])
    hdr = parse_fasta_headers(open(os.path.join(d, f"expect.{prefix}.reads.fasta")).read())
    assert [(names[r], b, e) for r, b, e in zip(got["frag_read"], got["frag_begin"], got["frag_end"])] == [(h[1], h[2], h[3]) for h in hdr]


def test_oracle_g3_symmetric_flip():
    """G3: the flag flips at record 4; target sides pushed before the flip must not count (repeat.hpp:54)."""
    d = os.path.join(GOLDEN, "micro", "g3")
    names, lens, cols = _micro_columns(d)
    p = _params_from_args(MAN["micro"]["g3"]["args"])
    got = oracle_run(p, lens, *cols)
    assert got["symmetric"] == 1
    cov_rows = parse_coverage_txt(open(os.path.join(d, "expect.raft.coverage.txt")).read())
    assert np.array_equal(got["cov"], np.concatenate(cov_rows))
    assert list(zip(got["rep_s"].tolist(), got["rep_e"].tolist())) == [(270, 630), (70, 430)]
    assert got["frag_begin"].tolist() == [0, 180, 780, 980, 1180, 0, 580, 780]  # chop.hpp:297-309 offsets
    assert got["frag_end"].tolist() == [200, 800, 1000, 1200, 1234, 600, 800, 900]


def test_oracle_defined_errors():
    p = RaftParams(est_cov=2, reso=50, repeat_length=100, interval_length=100, read_length=200, overlap_length=20)
    rl = np.array([230, 100], np.int32)
    one = lambda *v: [np.array([x], np.int32) for x in v]
    with pytest.raises(OracleError) as e:
        oracle_run(p, rl, *one(0, 0, 10, 2, 0, 10))        # unknown target id
    assert e.value.code == 2
    with pytest.raises(OracleError) as e:
        oracle_run(p, rl, *one(0, 0, 251, 1, 0, 10))       # reaches window 5 of a 5-window read
    assert e.value.code == 3
    oracle_run(p, rl, *one(0, 0, 250, 1, 0, 10))           # e > len but inside the last window: defined
    with pytest.raises(OracleError) as e:
        oracle_run(RaftParams(est_cov=2, read_length=100, interval_length=200, repeat_length=200), rl, *one(0, 0, 10, 1, 0, 10))
    assert e.value.code == 1                                # div == 0 (SIGFPE in the reference)
    with pytest.raises(OracleError) as e:                   # -v larger than the first kept cut point
        oracle_run(RaftParams(est_cov=2, reso=50, repeat_length=100, interval_length=100, read_length=200, overlap_length=250),
                   rl, *one(0, 0, 10, 1, 0, 10))
    assert e.value.code == 4


@pytest.mark.skipif(not have_ref_lib(), reason="oracle/_ref/libraft_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(600))
def test_oracle_vs_reference_code_fuzz(seed):
    """Differential fuzz against profileCoverage/repeat_annotate of the unmodified reference, in-process."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 40))
    reso = int(rng.choice([1, 7, 50, 64]))
    rl = rng.integers(0, 3000, n).astype(np.int32)
    m = int(rng.integers(1, 400))
    qid = rng.integers(0, n, m).astype(np.int32)
    tid = rng.integers(0, n, m).astype(np.int32)

    def coords(ids):
        ln = rl[ids].astype(np.int64)
        nb = (ln + reso - 1) // reso
        hi = nb * reso                                   # any end <= nb*reso stays inside the last window
        a = (rng.random(m) * (hi + 1)).astype(np.int64)
        b = (rng.random(m) * (hi + 1)).astype(np.int64)
        s, e = np.minimum(a, b), np.maximum(a, b)
        inv = rng.random(m) < 0.1                        # some inverted / empty intervals
        s2 = np.where(inv, e, s); e2 = np.where(inv, s, e)
        return s2.astype(np.int32), e2.astype(np.int32)

    qs, qe = coords(qid)
    ts, te = coords(tid)
    if seed % 2 == 0 and m > 3:                          # plant the mirror of record 0 -> symmetric flip mid-stream
        k = int(rng.integers(1, m))
        qid[k], tid[k], qs[k], qe[k], ts[k], te[k] = tid[0], qid[0], ts[0], te[0], qs[0], qe[0]
    L = int(rng.choice([100, 250, 1000]))
    p = RaftParams(reso=reso, est_cov=int(rng.integers(1, 6)), cov_mul=float(rng.choice([1.0, 1.3, 1.5, 2.0])),
                   repeat_length=L, interval_length=L, read_length=L * int(rng.integers(1, 4)) + int(rng.integers(0, L)),
                   overlap_length=int(rng.integers(0, min(L, 100))), flanking_length=int(rng.integers(0, 300)))
    got = oracle_run(p, rl, qid, qs, qe, tid, ts, te)
    ref = ref_lib_run(p, rl, qid, qs, qe, tid, ts, te)
    assert got["symmetric"] == ref["symmetric"]
    for k in ("cov", "rep_offset", "rep_s", "rep_e"):
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.skipif(not have_ref_lib(), reason="oracle/_ref/libraft_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(150))
def test_oracle_repeat_sort_ties_vs_reference_code(seed):
    """repeat.hpp:170 with tied starts and > 16 repeats: the oracle's restated libstdc++ introsort against the real
    std::sort inside the unmodified repeat_annotate()."""
    p, cols = tie_case(seed)
    got = oracle_run(p, *cols)
    ref = ref_lib_run(p, *cols)
    for k in ("cov", "rep_offset", "rep_s", "rep_e"):
        assert np.array_equal(got[k], ref[k]), k


def test_tie_cases_do_permute():
    """The tie fixture is not vacuous: in some seeds the order differs from the stable (emission) order."""
    permuted = 0
    for seed in range(150):
        p, cols = tie_case(seed)
        got = oracle_run(p, *cols)
        for r in range(len(cols[0])):
            e = got["rep_e"][got["rep_offset"][r]:got["rep_offset"][r + 1]]
            s = got["rep_s"][got["rep_offset"][r]:got["rep_offset"][r + 1]]
            assert np.all(np.diff(s) >= 0)
            if np.any(np.diff(e) < 0):
                permuted += 1
    assert permuted >= 10, permuted


def test_oracle_vs_reference_binary_fuzz():
    """tests/golden/ref_fuzz.npz: 320 random text inputs run through the unmodified reference BINARY (make_ref_fuzz.py):
    coverage, repeats (incl. the std::sort tie corner), fragment bounds / read_num and the stdout statistics."""
    from raft_testlib import assert_matches_ref_fuzz, ref_fuzz_case, ref_fuzz_count
    assert ref_fuzz_count() >= 200
    for i in range(ref_fuzz_count()):
        p, cols, exp = ref_fuzz_case(i)
        assert_matches_ref_fuzz(oracle_run(p, *cols), exp, p, f"ref_fuzz case {i}")


def test_oracle_config1_standin():
    """BASELINE configs[0] restated: 2 Mbp / 42x / -e 42; expected arrays parsed from the files the
    reference binary wrote for the gz inputs."""
    from raft_testlib import load_config1
    p, cols, exp, meta = load_config1()
    got = oracle_run(p, *cols)
    assert got["symmetric"] == meta["symmetric"] == 1 and len(cols[1]) == meta["n_rec"]
    for k in EXP_KEYS:
        assert np.array_equal(got[k], exp[k]), k
    assert "coverage per window is %f \n" % (got["total_coverage"] / got["total_windows"]) in meta["stdout"]
    assert "fraction_of_repeat_length %f \n" % (got["total_repeat_length"] / got["total_read_length"]) in meta["stdout"]
