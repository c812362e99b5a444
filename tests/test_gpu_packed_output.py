"""GPU: passes that write the transfer encoding of cov[] directly (raft_hip_set_output_width, pileup_wave.hpp OW = 1 / 2).

The encoding is a lossless restatement of what repeat.hpp:102-108 hands its formatter, so every check is exact: the
decoded array equals the oracle's cov, every other output is untouched by the width, and the widths agree with each other.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
from raft_testlib import RaftParams, assert_same_result, load_config1, oracle_run, ref_fuzz_case, ref_fuzz_count

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def decode(codes, exc_index, exc_value):
    cov = codes.astype(np.int32)
    cov[exc_index] = exc_value
    return cov


def run_width(p, cols, width, variant="wave", force_bucket=False):
    """Everything a caller can see of one pass in the given output width (variant: raft_testlib.KERNELS)."""
    from raft_amd import engine
    from raft_testlib import kernel_mode
    eng = engine.Engine(p, device=0)
    try:
        eng.set_tuning(0, force_bucket, -1)
        eng.set_output_width(width)
        with kernel_mode(variant):
            eng.run_host(*cols)
            s = eng.finish()
        pk = eng.packed_device()
        res = {"summary": s, "packed": None if pk is None else {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in pk.items()}}
        if pk is not None and pk["width"] == 2:
            res["packed"]["cov8"] = res["packed"]["cov8"].view(np.uint16)      # (handed out as int16 bits)
        res["same"] = eng.fetch_packed(width=width if width != 4 else 1)
        res["other"] = eng.fetch_packed(width=2 if width == 1 else 1)
        got = eng.fetch()
        got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                   total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
        res["fetch"] = got
        return res
    finally:
        eng.close()


def check_against(res, want, width, what):
    assert_same_result(res["fetch"], want, what)
    limit = 255 if width == 1 else 65535
    if width != 4 and res["packed"] is not None:
        pk = res["packed"]
        assert pk["width"] == width, what
        assert np.array_equal(pk["cov8"], np.minimum(want["cov"], limit).astype(pk["cov8"].dtype)), what
        order = np.argsort(pk["exc_index"], kind="stable")
        big = np.flatnonzero(want["cov"] >= limit)
        assert np.array_equal(pk["exc_index"][order], big) and np.array_equal(pk["exc_value"][order], want["cov"][big]), what
    for key, w in (("same", width if width != 4 else 1), ("other", 2 if width == 1 else 1)):
        f = res[key]
        assert np.array_equal(decode(f["cov8"], f["exc_index"], f["exc_value"]), want["cov"]), (what, key)
        assert f["cov8"].dtype == (np.uint8 if w == 1 else np.uint16)
        assert np.all(np.diff(f["exc_index"]) > 0), (what, key)          # handed out ascending by window
        for k in ("rep_s", "rep_e", "frag_begin", "frag_end", "rep_offset", "frag_offset", "cov_offset"):
            assert np.array_equal(f[k], want[k]), (what, key, k)


@pytest.mark.parametrize("width", [1, 2])
@pytest.mark.parametrize("mode", ["auto", "bucket", "deep"])
def test_config1_in_every_width(width, mode):
    p, cols, exp, meta = load_config1()
    res = run_width(p, cols, width, variant="deep" if mode == "deep" else "wave", force_bucket=(mode == "bucket"))
    for k in exp:
        assert np.array_equal(res["fetch"][k], exp[k]), (mode, k)
    assert res["packed"] is not None                  # (both kernels write the encoding the context asked for)
    if res["packed"] is not None:
        assert np.array_equal(res["packed"]["cov8"], np.minimum(exp["cov"], 255 if width == 1 else 65535))


@pytest.mark.parametrize("width", [1, 2])
def test_reference_fuzz_in_every_width(width):
    from raft_testlib import assert_matches_ref_fuzz
    for i in range(0, ref_fuzz_count(), 3):
        p, cols, exp = ref_fuzz_case(i)
        res = run_width(p, cols, width)
        assert_matches_ref_fuzz(res["fetch"], exp, p, f"case {i} width {width}")


def _deep_set(seed, n_reads=300, depth=400):
    """Short reads piled `depth` deep over most of their length: nearly every window is at or above a byte's limit."""
    rng = np.random.default_rng(seed)
    rl = rng.integers(20_000, 60_000, n_reads).astype(np.int32)
    qid = np.repeat(np.arange(n_reads, dtype=np.int32), depth)
    a = (rng.random(qid.size) * 0.1 * rl[qid]).astype(np.int32)
    b = (rl[qid] - rng.random(qid.size) * 0.1 * rl[qid]).astype(np.int32)
    b = np.maximum(b, a + 1)
    return rl, qid, a, b


@pytest.mark.parametrize("width", [1, 2, 4])
def test_most_windows_above_a_byte(width):
    """Width 1 on a 400-deep set: the exception list overflows its first size, the pass is run again with room; widths 2
    and 4 never notice.  All equal the oracle."""
    rl, qid, a, b = _deep_set(5)
    p = RaftParams(est_cov=200, symmetric_mode=1)        # (high_cov = 300: the plateau of every read is a repeat)
    want = oracle_run(RaftParams(**dict(p.__dict__, symmetric_mode=-1)), rl, qid, a, b, qid, a, b)
    want["symmetric"] = 1
    assert (want["cov"] >= 255).mean() > 0.5 and want["rep_s"].size > 0
    res = run_width(p, (rl, qid, a, b, None, None, None), width)
    check_against(res, want, width, f"deep width {width}")
    if width == 1:
        assert res["packed"]["exc_index"].size == int((want["cov"] >= 255).sum()) > 4096


@pytest.mark.parametrize("width", [1, 2])
@pytest.mark.parametrize("seed", [1, 2])
def test_long_reads_and_pieces_in_every_width(width, seed):
    """Extra tiles (groups, pieces of reads longer than the LDS window) write the encoding too; tile edges fall on any byte."""
    from test_gpu_configs import _long_read_set
    rl, qid, a, b = _long_read_set(seed)
    for p in (RaftParams(est_cov=14, symmetric_mode=1), RaftParams(est_cov=9, reso=37, repeat_length=3000, interval_length=3000,
                                                                  read_length=9000, flanking_length=5000, symmetric_mode=1)):
        want = oracle_run(RaftParams(**dict(p.__dict__, symmetric_mode=-1)), rl, qid, a, b, qid, a, b)
        want["symmetric"] = 1
        res = run_width(p, (rl, qid, a, b, None, None, None), width)
        assert res["packed"] is not None
        check_against(res, want, width, f"long reads seed {seed} width {width} reso {p.reso}")


@pytest.mark.parametrize("width", [1, 2, 8])
def test_existing_suites_with_every_context_in_that_width(width):
    """RAFT_COV_WIDTH puts every context of a process into the width (8: the four-bit step encoding): the parity and
    consistency suites, whose checks all go through the int32 array, must not notice."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_consistency.py")],
                       cwd=ROOT, env=dict(os.environ, RAFT_COV_WIDTH=str(width)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail
