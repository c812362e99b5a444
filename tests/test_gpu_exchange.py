"""GPU: the native exchange step of a pre-split PAF (BASELINE configs[3]; raft_hip_exchange / raft_hip_exchange_local) and
grouped input of more runs than the pileup kernels take (merged on the device).

One box, one GPU: `raft_hip_exchange_local` runs every rank as a context of this process (peer copies degenerate to
device-to-device copies); the RCCL form runs with a one-rank communicator -- RCCL cannot put two ranks on one GPU, so more
ranks over RCCL are the multi-GPU node's to run (bench.py --presplit).  Either way: what every rank receives, piled up by
its engine, equals the oracle's outputs for that rank's reads."""
import numpy as np
import pytest
from raft_testlib import RaftParams, assert_same_result, oracle_run

pytestmark = pytest.mark.gpu


def rank_slice_of(want, b0, b1):
    """The oracle's outputs restricted to reads [b0, b1), offsets rebased."""
    out = {}
    for key, arrs in (("cov", ("cov",)), ("rep", ("rep_s", "rep_e")), ("cut", ("cuts",)), ("frag", ("frag_read", "frag_begin", "frag_end"))):
        off = want[key + "_offset"]
        out[key + "_offset"] = off[b0:b1 + 1] - off[b0]
        for k in arrs:
            out[k] = want[k][off[b0]:off[b1]]
    out["frag_read"] = out["frag_read"] - b0
    return out


def make_slices(torch, hostio, engine, o, world, windows=False):
    """The record stream cut into `world` contiguous slices (what each rank's tokeniser would hold), each in grouped form;
    `windows`: the records as window records (one word each) instead of the two coordinate columns."""
    n_rec = o.n_rec
    slices = []
    for g in range(world):
        lo, hi = n_rec * g // world, n_rec * (g + 1) // world
        off = hostio.group_offsets(o.n_reads, o.qid[lo:hi].numpy(), max_runs=4)
        assert off is not None and off.shape[0] <= 2
        if windows:
            w = hostio.pack_windows(o.qs[lo:hi].numpy(), o.qe[lo:hi].numpy(), 50)
            slices.append(engine.Slice(off, torch.as_tensor(w.view(np.int32)).to("cuda:0")))
        else:
            slices.append(engine.Slice(off, o.qs[lo:hi].to("cuda:0").contiguous(), o.qe[lo:hi].to("cuda:0").contiguous()))
    return slices


def check_rank(torch, eng, p, got, read_len, b0, b1, want, what):
    rl = read_len[b0:b1]
    d_rl = torch.as_tensor(np.ascontiguousarray(rl)).to("cuda:0")
    B = int(((rl.astype(np.int64) + p.reso - 1) // p.reso).sum())
    if got["qe"] is None:            # window records arrived
        eng.run_device_windows(d_rl, got["rec_offset"], got["qs"], n_bins=B)
    else:
        eng.run_device_grouped(d_rl, got["rec_offset"], None, got["qs"], got["qe"], n_bins=B)
    s = eng.finish()
    res = eng.fetch()
    exp = rank_slice_of(want, b0, b1)
    for k in exp:
        assert np.array_equal(res[k], exp[k]), (what, k)
    assert s.interval_path == 0 and s.n_segments == got["n_runs"] and s.n_intervals == got["n_rec"]
    return s


@pytest.mark.parametrize("windows", [False, True])
@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_exchange_local_then_pass_equals_oracle(world, windows):
    import torch
    from raft_amd import dist as rdist
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=5000, seed=31 + world)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    engs = [engine.Engine(RaftParams(est_cov=30, symmetric_mode=1), device=0) for _ in range(world)]
    slices = make_slices(torch, hostio, engine, o, world, windows)
    ipr = torch.bincount(o.qid.long(), minlength=o.n_reads)
    bounds = rdist.partition_reads(o.read_len, p.reso, world, intervals_per_read=ipr).numpy()
    for rep in range(2):                                  # a second exchange reuses the contexts' buffers
        got = engine.exchange_local(engs, bounds, slices)
        assert sum(g["n_rec"] for g in got) == o.n_rec
        tot = 0
        for g in range(world):
            assert got[g]["n_runs"] <= 2 * world and got[g]["n_reads"] == bounds[g + 1] - bounds[g]
            s = check_rank(torch, engs[g], p, got[g], cols[0], int(bounds[g]), int(bounds[g + 1]), want, f"world {world} rank {g} pass {rep}")
            tot += s.n_fragments
        assert tot == len(want["frag_read"])
    for e in engs:
        e.close()


@pytest.mark.parametrize("windows", [False, True])
def test_exchange_rccl_one_rank_communicator(windows):
    """RCCL itself: communicator from a unique id, the all-gather of the piece sizes, grouped send / receive (to itself) on the
    context's stream, the rebasing kernel -- with the one rank this box can hold."""
    import torch
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=4000, seed=77)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    eng = engine.Engine(RaftParams(est_cov=30, symmetric_mode=1), device=0)
    uid = engine.Comm.unique_id()
    assert len(uid) == 128
    comm = engine.Comm(0, uid, 0, 1)
    sl = make_slices(torch, hostio, engine, o, 1, windows)[0]
    bounds = np.array([0, o.n_reads], np.int64)
    for rep in range(2):
        got = comm.exchange(eng, bounds, sl)
        assert got["n_rec"] == o.n_rec and got["n_runs"] == 2
        check_rank(torch, eng, p, got, cols[0], 0, o.n_reads, want, f"rccl pass {rep}")
    comm.close()
    eng.close()


@pytest.mark.parametrize("k_runs", [5, 9, 16])
def test_more_than_four_runs_are_merged_on_the_device(k_runs):
    """A PAF concatenated from many files / what a rank of an eight-rank pre-split job receives: up to 16 sorted runs.  The
    pass merges them into one (no histogram, no atomics: the offsets say where everything goes) and stays on the
    sorted-segment path."""
    import torch
    from raft_amd import engine, hostio
    rng = np.random.default_rng(900 + k_runs)
    rl = rng.integers(0, 60000, 2500).astype(np.int32)
    rl[rng.integers(0, len(rl), 50)] = 0
    rl[7] = 700_000                                       # a read longer than the LDS window: pieces
    ok = np.flatnonzero(rl > 0)
    qid = np.concatenate([np.sort(rng.choice(ok, int(rng.integers(1, 9000)))) for _ in range(k_runs)]).astype(np.int32)
    a = (rng.random(len(qid)) * rl[qid]).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(len(qid)) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
    p = RaftParams(est_cov=8)
    want = oracle_run(p, rl, qid, a, b, qid, a, b); want["symmetric"] = 1
    assert hostio.group_offsets(len(rl), qid, max_runs=4) is None
    off = hostio.group_offsets(len(rl), qid, max_runs=16)
    assert off is not None and off.shape[0] == k_runs
    eng = engine.Engine(RaftParams(est_cov=8, symmetric_mode=1), device=0)
    t = lambda x, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(x)).to(dt).to("cuda:0")
    B = int(((rl.astype(np.int64) + 49) // 50).sum())
    for q, hint in ((None, B), (t(qid), -1), (None, -1), (t(qid), B - 5)):
        eng.run_device_grouped(t(rl), t(off, torch.int64), q, t(a), t(b), n_bins=hint)
        s = eng.finish()
        got = eng.fetch()
        got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                   total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
        assert_same_result(got, want, f"{k_runs} runs, qid {q is not None}, hint {hint}")
        assert s.interval_path == 0 and s.n_segments == k_runs
    eng.run_host_grouped(rl, off, a, b)
    assert eng.finish().n_fragments == len(want["frag_read"])
    bad = off.copy(); bad[3, 100] = bad[3, 101] + 1       # offsets that step back are the caller's error, whatever the run count
    with pytest.raises(engine.RaftError) as e:
        eng.run_device_grouped(t(rl), t(bad, torch.int64), None, t(a), t(b), n_bins=B); eng.finish()
    assert e.value.code == engine.ERR_PARAM
    eng.close()
