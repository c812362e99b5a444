"""GPU: a pass built on what the context's previous pass found (round 6, engine.hip run_pass `speculate`).

raft_hip_run_device waits for the device once on its way -- sizes, sorted runs -- unless the context's last pass went the
sorted-run way over a stream of the same shape (counts, column addresses, parameters): then the pass is built on what that one
found and the device verifies it.  Whatever is in the buffers, the results are the oracle's (chop.hpp:133-322, repeat.hpp:28-171)."""
import numpy as np
import pytest
from raft_testlib import assert_same_result, oracle_run

from raft_amd.params import RaftParams

pytestmark = pytest.mark.gpu
SPECULATED = 2


def _set(seed, n_reads=3000, n=60000, two_runs=True):
    rng = np.random.default_rng(seed)
    rl = rng.integers(3000, 40000, n_reads).astype(np.int32)
    runs = []
    for k in range(2 if two_runs else 1):
        qid = np.sort(rng.integers(0, n_reads, n // 2)).astype(np.int32)
        a = (rng.random(qid.size) * rl[qid] * 0.8).astype(np.int32)
        b = np.minimum(rl[qid], a + 1 + (rng.random(qid.size) * rl[qid] * 0.3).astype(np.int32)).astype(np.int32)
        runs.append((qid, a, b))
    return rl, tuple(np.concatenate([r[i] for r in runs]) for i in range(3))


def _result(eng, s):
    got = eng.fetch()
    got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
               total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    return got


def test_second_pass_over_the_same_buffers_speculates_and_equals_the_oracle():
    import torch
    from raft_amd import engine
    p = RaftParams(est_cov=8, symmetric_mode=1)
    rl, (qid, a, b) = _set(1)
    want = oracle_run(p, rl, qid, a, b, qid, a, b); want["symmetric"] = 1
    dev = [torch.from_numpy(x).to("cuda:0") for x in (rl, qid, a, b)]
    eng = engine.Engine(p, device=0)
    flags = []
    for it in range(4):
        eng.run_device(*dev)
        s = eng.finish()
        flags.append(s.flags & SPECULATED)
        assert_same_result(_result(eng, s), want, f"pass {it}")
    assert flags[0] == 0 and all(f == SPECULATED for f in flags[1:]), flags
    # other coordinates in the same buffers: what was assumed (windows, run ends) still holds, the records are read afresh
    rng = np.random.default_rng(5)
    a2 = (rng.random(qid.size) * rl[qid] * 0.5).astype(np.int32)
    b2 = np.minimum(rl[qid], a2 + 1 + (rng.random(qid.size) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
    dev[2].copy_(torch.from_numpy(a2)); dev[3].copy_(torch.from_numpy(b2))
    want2 = oracle_run(p, rl, qid, a2, b2, qid, a2, b2); want2["symmetric"] = 1
    eng.run_device(*dev)
    s = eng.finish()
    assert s.flags & SPECULATED
    assert_same_result(_result(eng, s), want2, "other coordinates, same shape")
    eng.close()


@pytest.mark.parametrize("what", ["lengths", "same_windows", "run_ends", "unsorted", "bad_id"])
def test_a_stream_that_is_not_what_was_assumed_is_run_the_long_way(what):
    """Same counts, same addresses, other contents: the device refutes the assumption (other window count; other run ends; no
    sorted runs at all; an id out of range) and the result -- or the error -- is what a fresh context gives."""
    import torch
    from raft_amd import engine
    p = RaftParams(est_cov=8, symmetric_mode=1)
    rl, (qid, a, b) = _set(2)
    dev = [torch.from_numpy(x).to("cuda:0") for x in (rl, qid, a, b)]
    eng = engine.Engine(p, device=0)
    for _ in range(2):
        eng.run_device(*dev); s = eng.finish()
    assert s.flags & SPECULATED
    rng = np.random.default_rng(8)
    rl2, qid2, a2, b2 = rl.copy(), qid.copy(), a.copy(), b.copy()
    if what == "lengths":
        rl2 = rl + 50 * rng.integers(1, 4, rl.size).astype(np.int32)          # more windows per read, the records still fit
    elif what == "same_windows":
        # every read one base longer where that leaves its window count alone: the geometry the context holds would still be right for the
        # pileup, the fragments' would not (chop.hpp:209-223) -- the lengths themselves are compared, not what was derived from them
        rl2 = rl + ((rl % 50 > 0) & (rl % 50 < 49)).astype(np.int32)
        assert (rl2 != rl).any() and np.array_equal((rl2 + 49) // 50, (rl + 49) // 50)
    elif what == "run_ends":
        h = qid.size // 2                                                     # the second run begins 1000 records later
        q = np.concatenate([np.sort(qid[:h + 1000]), np.sort(qid[h + 1000:])]).astype(np.int32)
        qid2 = q
        a2 = (rng.random(q.size) * rl[q] * 0.5).astype(np.int32); b2 = np.minimum(rl[q], a2 + 100).astype(np.int32)
    elif what == "unsorted":
        perm = rng.permutation(qid.size)
        qid2, a2, b2 = qid[perm], a[perm], b[perm]
    else:
        qid2[qid.size // 3] = rl.size + 7
    for t, x in zip(dev, (rl2, qid2, a2, b2)):
        t.copy_(torch.from_numpy(np.ascontiguousarray(x)))
    if what == "bad_id":
        eng.run_device(*dev)
        with pytest.raises(engine.RaftError) as e:
            eng.finish()
        assert e.value.code == engine.ERR_READ_ID
    else:
        want = oracle_run(p, rl2, qid2, a2, b2, qid2, a2, b2); want["symmetric"] = 1
        eng.run_device(*dev)
        s = eng.finish()
        # (lengths that leave every window count alone refute the pass only where it kept the geometry it held: with
        # RAFT_NO_KEEP_GEOMETRY=1 the scan runs over the new lengths and the pass stands)
        assert what == "same_windows" or not (s.flags & SPECULATED)
        assert_same_result(_result(eng, s), want, what)
        # ... and the context is itself again afterwards
        eng.run_device(*dev); s = eng.finish()
        assert_same_result(_result(eng, s), want, what + ", next pass")
    eng.close()


def test_detecting_context_speculates_too_and_keeps_detecting():
    """symmetric_mode = -1: the mirror of record 0 is searched in every pass, speculative or not (chop.hpp:171-184)."""
    import torch
    from raft_amd import engine
    rng = np.random.default_rng(3)
    n_reads, n = 1500, 20000
    rl = rng.integers(3000, 40000, n_reads).astype(np.int32)
    qid = np.sort(rng.integers(0, n_reads, n)).astype(np.int32)
    tid = rng.integers(0, n_reads, n).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.8).astype(np.int32); b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.2).astype(np.int32)).astype(np.int32)
    ta = (rng.random(n) * rl[tid] * 0.8).astype(np.int32); tb = np.minimum(rl[tid], ta + 1 + (rng.random(n) * rl[tid] * 0.2).astype(np.int32)).astype(np.int32)
    order = np.argsort(tid, kind="stable")
    sym = tuple(np.concatenate([x, y[order]]) for x, y in ((qid, tid), (a, ta), (b, tb), (tid, qid), (ta, a), (tb, b)))
    p = RaftParams(est_cov=10)
    want = oracle_run(p, rl, *sym)
    assert want["symmetric"] == 1
    dev = [torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0") for x in (rl,) + sym]
    eng = engine.Engine(p, device=0)
    for it in range(3):
        eng.run_device(*dev); s = eng.finish()
        assert bool(s.flags & SPECULATED) == (it > 0)
        assert_same_result(_result(eng, s), want, f"pass {it}")
    # the mirror goes away (record 0's mirror gets another target start): same shape, not symmetric any more -- the target sides of the
    # second half (= the query sides of the first) are then piled up a second time, as the reference does (chop.hpp:165-169)
    mirror = int(np.flatnonzero((sym[0][n:] == sym[3][0]) & (sym[3][n:] == sym[0][0]) & (sym[4][n:] == sym[1][0]) & (sym[5][n:] == sym[2][0]))[0]) + n
    dev[5][mirror] = max(0, int(sym[4][mirror]) - 1) if int(sym[4][mirror]) > 0 else int(sym[4][mirror]) + 1
    cols = [t.cpu().numpy() for t in dev]
    want2 = oracle_run(p, *cols)
    assert want2["symmetric"] == 0
    eng.run_device(*dev); s = eng.finish()
    assert_same_result(_result(eng, s), want2, "mirror gone")
    eng.close()


def test_a_deep_pile_appears_in_a_stream_that_had_none():
    """A speculative pass over a stream whose last pass listed no deep tile does not launch the 32-bit side kernel.  Same counts, same
    run, same window count -- but now 40,000 records sit on one read: the wave kernel finds no room for the tile, the pass is run
    again the long way (RERUN), and the result is the oracle's; the passes after that launch the side kernel and need no re-run."""
    import torch
    from raft_amd import engine
    p = RaftParams(est_cov=8, symmetric_mode=1)
    rng = np.random.default_rng(21)
    n_reads, n = 1200, 48000
    rl = rng.integers(30000, 60000, n_reads).astype(np.int32)
    qid = np.repeat(np.arange(n_reads, dtype=np.int32), n // n_reads)               # 40 records per read, one sorted run
    a = (rng.random(n) * rl[qid] * 0.5).astype(np.int32); b = (a + 1 + (rng.random(n) * rl[qid] * 0.4).astype(np.int32)).astype(np.int32)
    dev = [torch.from_numpy(x).to("cuda:0") for x in (rl, qid, a, b)]
    eng = engine.Engine(p, device=0)
    for _ in range(3):
        eng.run_device(*dev); s = eng.finish()
    assert s.flags & SPECULATED and not (s.flags & 4)
    qid2 = np.sort(np.concatenate([qid[::6][:n - 40000], np.full(40000, 700, np.int32)])).astype(np.int32)      # 8,000 records spread over the reads + the pile
    a2 = (rng.random(n) * rl[qid2] * 0.5).astype(np.int32); b2 = (a2 + 1 + (rng.random(n) * rl[qid2] * 0.4).astype(np.int32)).astype(np.int32)
    want = oracle_run(p, rl, qid2, a2, b2, qid2, a2, b2); want["symmetric"] = 1
    assert want["cov"].max() >= 10000
    for t, x in zip(dev[1:], (qid2, a2, b2)):
        t.copy_(torch.from_numpy(x))
    flags = []
    for it in range(3):
        eng.run_device(*dev); s = eng.finish()
        flags.append(s.flags)
        assert_same_result(_result(eng, s), want, f"deep pile, pass {it}")
    assert flags[0] & 8 and flags[0] & 4 and not (flags[0] & SPECULATED), flags         # re-run, deep tiles taken
    assert all(f & 4 and not (f & 8) for f in flags[1:]), flags
    eng.close()


def test_another_set_of_reads_in_between_does_not_leave_its_geometry_behind():
    """A speculative pass keeps the per-read geometry the context holds when nobody has written it since the pass it is built on.  A
    grouped pass over OTHER reads writes it: the next pass over the first set -- same buffers, same shape, speculative -- must not run
    on the other set's window offsets."""
    import torch
    from raft_amd import engine, hostio
    p = RaftParams(est_cov=8, symmetric_mode=1)
    rl, (qid, a, b) = _set(11)
    want = oracle_run(p, rl, qid, a, b, qid, a, b); want["symmetric"] = 1
    dev = [torch.from_numpy(x).to("cuda:0") for x in (rl, qid, a, b)]
    rl_o, (qid_o, a_o, b_o) = _set(12, n_reads=3000, n=40000, two_runs=False)      # as many reads, other lengths
    want_o = oracle_run(p, rl_o, qid_o, a_o, b_o, qid_o, a_o, b_o); want_o["symmetric"] = 1
    off_o = hostio.group_offsets(rl_o.size, qid_o)
    eng = engine.Engine(p, device=0)
    for it in range(3):
        eng.run_device(*dev); s = eng.finish()
    assert s.flags & SPECULATED
    assert_same_result(_result(eng, s), want, "first set")
    eng.run_host_grouped(rl_o, off_o, a_o, b_o); s = eng.finish()
    assert_same_result(_result(eng, s), want_o, "the other set, grouped")
    for it in range(2):
        eng.run_device(*dev); s = eng.finish()
        assert_same_result(_result(eng, s), want, f"first set again, pass {it}")
    assert s.flags & SPECULATED
    eng.close()
