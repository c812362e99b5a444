"""GPU: tiles too deep for the wave kernel's 16-bit difference array (round 6: raft_amd/csrc/pileup_deep.hpp).

The reference has no depth limit (std::vector<int> counters, repeat.hpp:39-44, 62-77).  A wave tile with 2^15 intervals or more is
listed by pileup_wave_kernel and piled up by pileup_deep_kernel, 32-bit, a workgroup per tile, in the SAME pass -- until round 5
such a tile sent the whole pass to the int32 kernels of rounds 1-3.  Two kinds of tests: genuinely deep piles (coverage beyond
2^15: one pass, no re-run, every encoding), and the parity suites run again with the threshold lowered (RAFT_DEEP_MIN) so that
ordinary tiles -- several reads, pieces of long reads, every input form and encoding -- come through the deep kernel."""
import os
import subprocess
import sys

import numpy as np
import pytest
from raft_testlib import ROOT, assert_same_result, oracle_run

from raft_amd.params import RaftParams

pytestmark = pytest.mark.gpu
DEEP, RERUN = 4, 8


def _pile(m=40000, long_read=False, seed=9):
    rng = np.random.default_rng(seed)
    rl = np.array([30000, 12000, 400000 if long_read else 50000, 8000], np.int32)
    qid = np.concatenate([np.zeros(50, np.int32), np.full(m, 2, np.int32), np.full(30, 3, np.int32)])
    qs = np.zeros(qid.size, np.int32); qe = np.zeros(qid.size, np.int32)
    qs[50:50 + m] = rng.integers(0, 20000, m)
    qe[50:50 + m] = qs[50:50 + m] + (rng.integers(330000, 380000, m) if long_read else rng.integers(20000, 30000, m))
    qs[:50] = rng.integers(0, 10000, 50); qe[:50] = qs[:50] + 5000
    qs[50 + m:] = 100; qe[50 + m:] = 7000
    return rl, qid, qs, qe


@pytest.mark.parametrize("long_read", [False, True])
def test_a_deep_pile_is_one_pass(long_read):
    """40,000 intervals on one read (a tile of whole reads / the pieces of a read longer than a tile): coverage beyond 2^15 as the
    oracle has it, from ONE pass -- raft_hip_finish ran nothing again -- and the summary says the deep kernel took tiles."""
    from raft_amd import engine
    rl, qid, qs, qe = _pile(long_read=long_read)
    p = RaftParams(est_cov=30, symmetric_mode=1)
    want = oracle_run(p, rl, qid, qs, qe, qid, qs, qe); want["symmetric"] = 1
    assert want["cov"].max() >= 32768
    eng = engine.Engine(p, device=0)
    for it in range(2):
        eng.run_host(rl, qid, qs, qe, qid, qs, qe)
        s = eng.finish()
        assert s.flags & DEEP and not (s.flags & RERUN), s.flags
        got = eng.fetch()
        got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                   total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
        assert_same_result(got, want, f"deep pile, pass {it}")
    eng.close()


def test_more_deep_tiles_than_the_list_holds_grow_it(monkeypatch):
    """The list of deep tiles starts at 1024 entries; with the threshold at 1 every tile of a 3000-read set is listed, the pass is run
    again with room, and the context keeps the larger list."""
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    monkeypatch.setenv("RAFT_DEEP_MIN", "1")
    o = make_overlaps(12000, seed=4)
    p = RaftParams(est_cov=30)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    want = oracle_run(p, *cols)
    eng = engine.Engine(p, device=0)
    flags = []
    for it in range(2):
        eng.run_host(*cols)
        s = eng.finish()
        flags.append(s.flags)
        got = eng.fetch()
        got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                   total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
        assert_same_result(got, want, f"every tile deep, pass {it}")
    assert flags[0] & DEEP and flags[0] & RERUN and not (flags[1] & RERUN), flags
    eng.close()


@pytest.mark.parametrize("suite", ["test_gpu_parity.py", "test_gpu_wave.py test_gpu_windows.py", "test_gpu_delta4.py test_gpu_packed_output.py",
                                   "test_gpu_configs.py -k 'not full_size'", "test_gpu_grouped.py test_gpu_routed.py"])
def test_parity_suites_through_the_deep_kernel(suite):
    """RAFT_DEEP_MIN=40: nearly every tile of the suites' sets goes through pileup_deep_kernel -- tiles of many reads, pieces of long
    reads, coordinate columns and window records in, every coverage encoding out.  (The tests that count passes or time kernels see
    the same pass structure: the deep kernel runs inside the pass.)"""
    env = dict(os.environ, RAFT_DEEP_MIN="40")
    cmd = f"{sys.executable} -m pytest -x -q -m gpu -p no:cacheprovider " + " ".join(os.path.join("tests", x) if x.endswith(".py") else x for x in suite.split(" "))
    r = subprocess.run(cmd, shell=True, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-3000:]
