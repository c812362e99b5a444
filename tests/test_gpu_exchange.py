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


@pytest.mark.parametrize("shape", ["nonsym", "shuffled_sym"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_non_symmetric_presplit_group_sides_then_exchange_equals_oracle(world, shape):
    """A pre-split PAF that is NOT symmetric (target sides are piled up too, chop.hpp:165-169) or not sorted by query: every
    rank expands and sorts ITS slice's sides on the device (raft_hip_group_sides, ABI 9) -- one run sorted by read id, hence by
    owner -- and the exchange routes it like any grouped slice; each rank's grouped pass over what arrives equals the oracle's
    outputs for its reads.  No torch sort, no torch collective (raft_amd/dist.py's exchange_intervals did both)."""
    import torch
    from raft_amd import dist as rdist
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    sym = shape != "nonsym"
    o = make_overlaps(n_reads=4000, seed=51 + world, symmetric=sym, shuffle=True)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    assert want["symmetric"] == (1 if sym else 0)
    engs = [engine.Engine(RaftParams(est_cov=30, symmetric_mode=1), device=0) for _ in range(world)]
    weight = torch.bincount(o.qid.long(), minlength=o.n_reads)
    if not sym:
        weight = weight + torch.bincount(o.tid.long(), minlength=o.n_reads)
    bounds = rdist.partition_reads(o.read_len, p.reso, world, intervals_per_read=weight).numpy()
    slices, n_iv = [], 0
    for g in range(world):
        lo, hi = o.n_rec * g // world, o.n_rec * (g + 1) // world
        dcols = [c[lo:hi].to("cuda:0").contiguous() for c in o.columns()]
        sl = engs[g].group_sides(o.n_reads, *dcols, symmetric=sym)
        # the slice is one run sorted by read id: offsets ascend, chain from 0 to the number of intervals
        assert sl.off.shape == (1, o.n_reads + 1) and sl.off[0, 0] == 0 and (np.diff(sl.off[0]) >= 0).all() and sl.off[0, -1] == sl.qs.numel()
        expect = (hi - lo) if sym else (hi - lo) + int((o.qid[lo:hi] != o.tid[lo:hi]).sum())
        assert sl.qs.numel() == expect
        n_iv += expect
        slices.append(sl)
    got = engine.exchange_local(engs, bounds, slices)
    assert sum(g["n_rec"] for g in got) == n_iv
    tot = 0
    for g in range(world):
        s = check_rank(torch, engs[g], p, got[g], cols[0], int(bounds[g]), int(bounds[g + 1]), want, f"{shape} world {world} rank {g}")
        tot += s.n_fragments
    assert tot == len(want["frag_read"])
    # a read id out of range is reported, not sorted
    bad = o.qid[:100].clone(); bad[7] = o.n_reads + 3
    with pytest.raises(engine.RaftError) as e:
        engs[0].group_sides(o.n_reads, bad.to("cuda:0"), *[c[:100].to("cuda:0").contiguous() for c in o.columns()[1:]], symmetric=sym)
    assert e.value.code == engine.ERR_READ_ID
    empty = [torch.empty(0, dtype=torch.int32, device="cuda:0")] * 6
    sl0 = engs[0].group_sides(o.n_reads, *empty, symmetric=False)
    assert sl0.qs.numel() == 0 and not sl0.off.any()
    for e_ in engs:
        e_.close()


@pytest.mark.parametrize("world", [1, 3, 8])
def test_presplit_symmetric_flag_is_the_or_over_the_ranks(world):
    """chop.hpp:171-184 across ranks (raft_hip_presplit_symmetric_local / the RCCL form with one rank): record 0 is rank 0's first
    record, its mirror may lie in any rank's slice -- or nowhere."""
    import torch
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    engs = [engine.Engine(RaftParams(est_cov=30), device=0) for _ in range(world)]
    for sym in (True, False):
        o = make_overlaps(n_reads=3000, seed=5 + world, symmetric=sym, shuffle=True)
        cols = [c.to("cuda:0") for c in o.columns()]
        want = oracle_run(RaftParams(est_cov=30), *[c.numpy() for c in (o.read_len,) + o.columns()])["symmetric"]
        assert want == (1 if sym else 0)
        cut = [o.n_rec * g // world for g in range(world + 1)]
        parts = [[c[cut[g]:cut[g + 1]].contiguous() for c in cols] for g in range(world)]
        assert engine.presplit_symmetric_local(engs, parts) == sym
        if sym and world > 1:
            # the mirror of record 0 moved to the last rank / removed altogether
            q0, qs0, qe0, t0, ts0, te0 = (int(c[0]) for c in cols)
            m = (cols[0] == t0) & (cols[3] == q0) & (cols[4] == qs0) & (cols[5] == qe0) & (cols[1] == ts0) & (cols[2] == te0)
            m[0] = False
            keep = ~m
            rest = [c[keep] for c in cols]
            mirror = [c[m][:1] for c in cols]
            cut2 = [rest[0].numel() * g // world for g in range(world + 1)]
            parts2 = [[c[cut2[g]:cut2[g + 1]].contiguous() for c in rest] for g in range(world)]
            assert engine.presplit_symmetric_local(engs, parts2) is False
            parts2[-1] = [torch.cat([a, b]).contiguous() for a, b in zip(parts2[-1], mirror)]
            assert engine.presplit_symmetric_local(engs, parts2) is True
    empty = [[torch.empty(0, dtype=torch.int32, device="cuda:0")] * 6 for _ in range(world)]
    assert engine.presplit_symmetric_local(engs, empty) is False
    if world == 1:
        comm = engine.Comm(0, engine.Comm.unique_id(), 0, 1)
        for sym in (True, False):
            o = make_overlaps(n_reads=2000, seed=9, symmetric=sym)
            assert comm.symmetric(engs[0], [c.to("cuda:0") for c in o.columns()]) == sym
        comm.close()
    for e in engs:
        e.close()


def test_non_symmetric_presplit_at_scale_eight_ranks_equal_the_single_pass():
    """An eighth of BASELINE configs[2] made NON-symmetric (412 k reads, 3.6e7 records, 7e7 intervals), pre-split over eight ranks
    (contexts of this process): raft_hip_presplit_symmetric_local says "not symmetric", every rank expands and sorts its slice
    (raft_hip_group_sides), the exchange routes the runs, every rank's grouped pass equals the single non-symmetric pass of one
    engine over the whole set -- every array, read range by read range -- and the totals add up."""
    import torch
    from raft_amd import dist as rdist
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    world = 8
    o = make_overlaps(412_500, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0", symmetric=False)
    p = RaftParams(est_cov=32)
    one = engine.Engine(p, device=0)
    one.run_device(o.read_len, *o.columns())
    s = one.finish()
    assert s.symmetric == 0
    ref = {k: v.clone() for k, v in one.outputs_device().items()}
    one.close()
    engs = [engine.Engine(RaftParams(est_cov=32, symmetric_mode=1), device=0) for _ in range(world)]
    cut = [o.n_rec * g // world for g in range(world + 1)]
    parts = [[c[cut[g]:cut[g + 1]].contiguous() for c in o.columns()] for g in range(world)]
    assert engine.presplit_symmetric_local(engs, parts) is False
    weight = torch.bincount(o.qid.long(), minlength=o.n_reads) + torch.bincount(o.tid.long(), minlength=o.n_reads)
    bounds = rdist.partition_reads(o.read_len.cpu(), p.reso, world, intervals_per_read=weight.cpu()).numpy()
    slices = [engs[g].group_sides(o.n_reads, *parts[g], symmetric=False) for g in range(world)]
    assert sum(sl.qs.numel() for sl in slices) == s.n_intervals
    got = engine.exchange_local(engs, bounds, slices)
    tot = [0, 0, 0, 0]
    for g in range(world):
        b0, b1 = int(bounds[g]), int(bounds[g + 1])
        rl = o.read_len[b0:b1].contiguous()
        B = int(((rl.long() + p.reso - 1) // p.reso).sum())
        engs[g].run_device_grouped(rl, got[g]["rec_offset"], None, got[g]["qs"], got[g]["qe"], n_bins=B)
        sg = engs[g].finish()
        out = engs[g].outputs_device()
        for key, offk in (("cov", "cov_offset"), ("rep_s", "rep_offset"), ("rep_e", "rep_offset"), ("cuts", "cut_offset"),
                          ("frag_begin", "frag_offset"), ("frag_end", "frag_offset")):
            lo, hi = int(ref[offk][b0]), int(ref[offk][b1])
            assert torch.equal(out[key], ref[key][lo:hi]), (g, key)
        tot = [tot[0] + sg.n_fragments, tot[1] + sg.n_repeats, tot[2] + sg.total_coverage, tot[3] + sg.total_repeat_length]
    assert tot == [s.n_fragments, s.n_repeats, s.total_coverage, s.total_repeat_length]
    for e in engs:
        e.close()


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("shape", ["symmetric", "nonsym_shuffled"])
def test_presplit_job_of_one_process_equals_the_oracle(world, shape):
    """raft_hip_run_presplit_local (what `raft` runs with RAFT_RANKS=N): the stream cut into `world` slices, flag, sides, ONE
    exchange, every rank's grouped pass, outputs in read order -- all of it equal to the oracle's outputs for the whole set,
    in both widths of the coverage encoding."""
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    o = make_overlaps(3000, seed=31 + world)
    cols = [c.numpy().copy() for c in (o.read_len,) + o.columns()]
    if shape == "nonsym_shuffled":           # one record per pair, in random order: target sides are piled up too (chop.hpp:165-169)
        rng = np.random.default_rng(5)
        keep = np.flatnonzero(cols[1] < cols[4])
        keep = keep[rng.permutation(keep.size)]
        cols = [cols[0]] + [c[keep] for c in cols[1:]]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    assert want["symmetric"] == (1 if shape == "symmetric" else 0)
    for width in (1, 2):
        engs = [engine.Engine(p, device=0) for _ in range(world)]
        out = engs[0].host_output_buffers(cols[0], pinned=False, width=width)
        res, s = engs[0].run_presplit(*cols, others=engs[1:], out=out)
        cov = res["cov8"].astype(np.int32)
        cov[res["exc_index"]] = res["exc_value"]
        assert np.all(np.diff(res["exc_index"]) > 0)
        assert np.array_equal(cov, want["cov"]), (world, shape, width)
        for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
            assert np.array_equal(res[k], want[k]), (world, shape, width, k)
        assert s.symmetric == want["symmetric"] and s.high_cov == want["high_cov"] and s.n_records == cols[1].size
        for k in ("total_coverage", "total_windows", "total_repeat_length", "total_read_length"):
            assert getattr(s, k) == want[k], k
        assert s.n_fragments == want["frag_begin"].size and s.n_devices_used == world
        for e in engs:
            e.close()
