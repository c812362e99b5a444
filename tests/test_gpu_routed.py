"""GPU: record streams that are not a handful of runs sorted by query id (shuffled, non-symmetric, many files) through the
host-to-host entry points on several contexts: the host's threads bucket the intervals by read (create_pileup's job,
chop.hpp:155-169) and route consecutive read ranges to the contexts in turn (engine.hip run_routed, SURVEY.md §8e
host-routed mode).  Same outputs as the one-piece pass and the oracle; also what lifts the 2^29-records-per-pass limit."""
import numpy as np
import pytest
from raft_testlib import RaftParams, oracle_run
from test_gpu_grouped import check_pipelined
from test_gpu_parity import load_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ctx,n_chunks", [(1, 4), (2, 0), (2, 5), (3, 9)])
@pytest.mark.parametrize("name,mode", [("s300_nonsym_shuffled", -1), ("s300_nonsym_shuffled", 0), ("s300_sym_shuffled", -1),
                                       ("s300_sym_shuffled", 1), ("edge_reads", -1), ("s200_smallparams", -1)])
def test_routed_equals_oracle(name, mode, n_ctx, n_chunks):
    from raft_amd import engine
    p, cols, exp, meta = load_case(name)
    want = oracle_run(p, *cols)
    if mode == 1 and want["symmetric"] != 1:
        pytest.skip("not a symmetric PAF")
    pm = RaftParams(**dict(p.__dict__, symmetric_mode=mode))
    eng = engine.Engine(pm, device=0)
    others = [engine.Engine(RaftParams(est_cov=3, reso=7), device=0) for _ in range(n_ctx - 1)]
    out = eng.host_output_buffers(cols[0], pinned=False, width=2)
    for rep in range(2):
        res, s = eng.run_pipelined(*cols, n_chunks=n_chunks, out=out, others=others)
        from raft_amd import hostio
        assert np.array_equal(hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"]), want["cov"]), (name, rep)
        check_pipelined(dict(res, cov8=res["cov8"]), s, want, f"{name} mode {mode} ctx {n_ctx} chunks {n_chunks} pass {rep}")
        if n_ctx > 1 and n_chunks > 1 and len(cols[0]) > n_chunks and "shuffled" in name:
            assert s.n_devices_used == n_ctx, s.n_devices_used
    # the contexts are themselves again afterwards
    eng.run_host(*cols)
    assert eng.finish().symmetric == want["symmetric"]
    for e in [eng] + others:
        e.close()


def test_routed_errors():
    from raft_amd import engine
    rng = np.random.default_rng(5)
    rl = rng.integers(3000, 40000, 800).astype(np.int32)
    n = 30000
    qid = rng.integers(0, len(rl), n).astype(np.int32); tid = rng.integers(0, len(rl), n).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.8).astype(np.int32); b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.2).astype(np.int32)).astype(np.int32)
    ta = (rng.random(n) * rl[tid] * 0.8).astype(np.int32); tb = np.minimum(rl[tid], ta + 1 + (rng.random(n) * rl[tid] * 0.2).astype(np.int32)).astype(np.int32)
    eng = engine.Engine(RaftParams(est_cov=10), device=0)
    other = engine.Engine(RaftParams(est_cov=10), device=0)
    want = oracle_run(RaftParams(est_cov=10), rl, qid, a, b, tid, ta, tb)
    res, s = eng.run_pipelined(rl, qid, a, b, tid, ta, tb, n_chunks=6, others=[other])
    check_pipelined(res, s, want, "random non-symmetric")
    assert s.n_devices_used == 2 and s.symmetric == 0
    t2 = tid.copy(); t2[777] = len(rl) + 3
    with pytest.raises(engine.RaftError) as e:
        eng.run_pipelined(rl, qid, a, b, t2, ta, tb, n_chunks=6, others=[other])
    assert e.value.code == engine.ERR_READ_ID and e.value.index == 777
    b2 = tb.copy(); b2[4242] = rl[tid[4242]] + 9000
    with pytest.raises(engine.RaftError) as e1:
        eng.run_pipelined(rl, qid, a, b, tid, ta, b2, n_chunks=6, others=[other])
    with pytest.raises(engine.RaftError) as e2:
        eng.run_host(rl, qid, a, b, tid, ta, b2); eng.finish()
    # (on the counting-sort path the index of a coordinate error counts the bucketed intervals, whose order inside a read is
    # not defined: include/raft_hip.h)
    assert e1.value.code == e2.value.code == engine.ERR_COORD
    eng.close(); other.close()


def test_routed_exception_list_too_small_reports_the_job_total():
    """A shuffled, non-symmetric stream whose windows at or above the byte limit outnumber exc_cap: raft_hip_run_multi returns
    RAFT_HIP_ERR_TOO_LARGE with the JOB's exception count in out->n_exc (include/raft_hip.h: one retry suffices), and the
    retry with that much room gives the oracle's arrays."""
    from raft_amd import engine, hostio
    rng = np.random.default_rng(11)
    rl = rng.integers(3000, 20000, 600).astype(np.int32)
    n = 400000                                          # ~600 intervals per read on ~230 windows: most windows are above 255
    qid = rng.integers(0, len(rl), n).astype(np.int32); tid = rng.integers(0, len(rl), n).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.5).astype(np.int32); b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
    ta = (rng.random(n) * rl[tid] * 0.5).astype(np.int32); tb = np.minimum(rl[tid], ta + 1 + (rng.random(n) * rl[tid] * 0.5).astype(np.int32)).astype(np.int32)
    p = RaftParams(est_cov=400, symmetric_mode=-1)
    want = oracle_run(p, rl, qid, a, b, tid, ta, tb)
    n_big = int((want["cov"] >= 255).sum())
    assert n_big > 5000
    eng = engine.Engine(p, device=0)
    other = engine.Engine(p, device=0)
    out = eng.host_output_buffers(rl, pinned=False, width=1, exc_cap=64)
    with pytest.raises(engine.RaftError) as e:
        eng.run_pipelined(rl, qid, a, b, tid, ta, tb, n_chunks=5, out=out, others=[other])
    assert e.value.code == engine.ERR_TOO_LARGE and eng.last_n_exc == n_big, (e.value.code, eng.last_n_exc, n_big)
    out = eng.host_output_buffers(rl, pinned=False, width=1, exc_cap=eng.last_n_exc)
    res, s = eng.run_pipelined(rl, qid, a, b, tid, ta, tb, n_chunks=5, out=out, others=[other])
    assert s.n_devices_used == 2 and res["exc_index"].size == n_big
    assert np.array_equal(hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"]), want["cov"])
    check_pipelined(res, s, want, "routed, exception list grown after TOO_LARGE")
    eng.close(); other.close()


def test_more_records_than_one_pass_takes():
    """5.5e8 non-symmetric records (> 2^29, the limit of a one-piece pass): refused as RAFT_HIP_ERR_TOO_LARGE before; now
    routed in read ranges.  Checked by invariants (the oracle does not finish at this size)."""
    import psutil
    if psutil.virtual_memory().available < 160 * (1 << 30):
        pytest.skip("needs ~60 GB of host memory with room to spare")
    from raft_amd import engine
    rng = np.random.default_rng(7)
    N, n = 300_000, 550_000_000
    rl = rng.integers(15000, 25000, N).astype(np.int32)
    p = RaftParams(est_cov=2000, reso=50)
    qid = rng.integers(0, N, n, dtype=np.int32); tid = rng.integers(0, N, n, dtype=np.int32)

    def coords(ids):
        ln = rl[ids]
        s = (rng.random(n, dtype=np.float32) * (ln - 2000)).astype(np.int32)
        return s, s + 1 + (rng.random(n, dtype=np.float32) * 1900).astype(np.int32)
    qs, qe = coords(qid)
    ts, te = coords(tid)
    eng = engine.Engine(p, device=0)
    other = engine.Engine(p, device=0)
    out = eng.host_output_buffers(rl, pinned=False, width=2, exc_cap=1 << 16)
    res, s = eng.run_pipelined(rl, qid, qs, qe, tid, ts, te, out=out, others=[other])
    differ = tid != qid
    touched = int(((qe.astype(np.int64) - 1) // 50 - qs // 50 + 1).sum()) + int((((te.astype(np.int64) - 1) // 50 - ts // 50 + 1) * differ).sum())
    assert s.n_records == n and s.n_intervals == n + int(differ.sum()) and s.symmetric == 0 and s.n_devices_used == 2
    assert s.total_coverage == touched == int(res["cov8"].astype(np.int64).sum())
    assert s.n_bins == int(((rl.astype(np.int64) + 49) // 50).sum()) and s.n_fragments >= N
    fo = res["frag_offset"]
    assert np.all(res["frag_begin"][fo[:-1]] == 0) and np.all(res["frag_end"][fo[1:] - 1] == rl)
    eng.close(); other.close()


@pytest.mark.parametrize("shape", ["fits", "wide", "errors"])
def test_large_shuffled_stream_buckets_window_records_or_coordinate_pairs(shape, monkeypatch):
    """The device pass over a million-odd records in random order (round 5: a hand-written LSD radix sort by read id,
    sort_pairs.hpp).  Where every window index fits 16 bits a side is ONE 64-bit item through the sort and the pileup kernel gets
    window records; `wide`: reads of 70,000 windows (-r 1) -- a side raises kErrWide and the pass is run again with coordinate
    pairs, for good on that context; `errors`: a negative coordinate / an interval past its read's end are reported by either
    route.  Outputs equal the oracle's on both routes (RAFT_NO_BUCKET_WINDOWS=1 forces the second)."""
    from raft_amd import engine
    from raft_testlib import assert_same_result
    rng = np.random.default_rng(23)
    wide = shape == "wide"
    p = RaftParams(est_cov=25, reso=1 if wide else 50, repeat_length=2000, interval_length=2000, read_length=6000, overlap_length=100, flanking_length=50)
    n_reads = 160 if wide else 9000
    rl = rng.integers(60000, 71000, n_reads).astype(np.int32) if wide else rng.integers(2000, 40000, n_reads).astype(np.int32)
    n = (1 << 20) + 12345                                     # (the sort takes over from the counting sort at 2^20 intervals)
    qid = rng.integers(0, n_reads, n).astype(np.int32); tid = rng.integers(0, n_reads, n).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.9).astype(np.int32); b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.1).astype(np.int32)).astype(np.int32)
    ta = (rng.random(n) * rl[tid] * 0.9).astype(np.int32); tb = np.minimum(rl[tid], ta + 1 + (rng.random(n) * rl[tid] * 0.1).astype(np.int32)).astype(np.int32)
    a[5] = b[5] = 77                                           # an empty, unaligned interval (still one window)
    cols = (rl, qid, a, b, tid, ta, tb)
    want = oracle_run(p, *cols)
    assert want["symmetric"] == 0
    for force_pairs in (False, True):
        if force_pairs:
            monkeypatch.setenv("RAFT_NO_BUCKET_WINDOWS", "1")
        eng = engine.Engine(p, device=0)
        for rep in range(2):                                   # (`wide`: the second pass goes straight to coordinate pairs)
            eng.run_host(*cols)
            s = eng.finish()
            got = eng.fetch()
            got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                       total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
            assert_same_result(got, want, f"{shape} pairs={force_pairs} pass {rep}")
            assert s.interval_path == 1 and s.n_intervals == int((qid != tid).sum()) + n
        if shape == "errors":
            b2 = tb.copy(); b2[4242] = rl[tid[4242]] + 9000
            eng.run_host(rl, qid, a, b, tid, ta, b2)
            with pytest.raises(engine.RaftError) as e1:
                eng.finish()
            assert e1.value.code == engine.ERR_COORD
            a2 = a.copy(); a2[999] = -3
            eng.run_host(rl, qid, a2, b, tid, ta, tb)
            with pytest.raises(engine.RaftError) as e2:
                eng.finish()
            assert e2.value.code == engine.ERR_COORD
            t2 = tid.copy(); t2[31337] = n_reads
            eng.run_host(rl, qid, a, b, t2, ta, tb)
            with pytest.raises(engine.RaftError) as e3:
                eng.finish()
            assert e3.value.code == engine.ERR_READ_ID and e3.value.index == 31337
        eng.close()
