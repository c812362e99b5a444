"""GPU: grouped input as WINDOW RECORDS (raft_hip_run_device_windows / run_host_windows / run_multi_windows) against the CPU
oracle, the golden outputs of the reference binary and the coordinate-column entries.

A window record is what profileCoverage uses of an interval (repeat.hpp:69-72: the windows qs / reso .. (qe - 1) / reso),
cut from the coordinates where they are tokenised (raft_host_pack_windows): one 32-bit word per record and no read id -- the
read is where the caller's offsets say.  The default configuration takes them in the pileup kernel itself (pileup_wave.hpp
IN = 1: the reads of a wave's records come from the tile's slice of the offsets); every other configuration, more than two
runs and the pass's fallbacks unpack them to coordinate columns first.  Bar: bit-exact.
"""
import json
import os

import numpy as np
import pytest
from raft_testlib import GOLDEN, RaftParams, assert_same_result, oracle_run

pytestmark = pytest.mark.gpu

MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def sym_params(p):
    return RaftParams(**dict(p.__dict__, symmetric_mode=1))


def n_windows(p, rl):
    return int(((np.asarray(rl, np.int64) + p.reso - 1) // p.reso).sum())


def result_of(eng, s):
    got = eng.fetch()
    got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
               total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    return got


def window_runs(eng, p, rl, off, win, want, what, widths=(4, 1, 2), with_host=True):
    """Every form of the windows entry on one context; each must equal `want`."""
    import torch
    dev = "cuda:0"
    d_rl = torch.as_tensor(np.ascontiguousarray(rl)).to(torch.int32).to(dev)
    d_off = torch.as_tensor(np.ascontiguousarray(off)).to(torch.int64).to(dev)
    d_w = torch.as_tensor(np.ascontiguousarray(win).view(np.int32)).to(dev)
    B = n_windows(p, rl)
    for width in widths:
        eng.set_output_width(width)
        for form, hint in (("hint", B), ("no hint", -1), ("wrong hint (low)", max(B - 17, 0)), ("wrong hint (high)", B + 4096)):
            eng.run_device_windows(d_rl, d_off, d_w, n_bins=hint)
            s = eng.finish()
            assert_same_result(result_of(eng, s), want, f"{what}: width {width}, {form}")
            assert s.interval_path == 0 and s.n_segments == off.shape[0] and s.n_bins == B
    eng.set_output_width(4)
    if with_host:
        eng.run_host_windows(rl, off, win)
        s = eng.finish()
        assert_same_result(result_of(eng, s), want, f"{what}: run_host_windows")


@pytest.mark.parametrize("name", sorted(n for n, m in MAN["synthetic"].items() if m["symmetric"] == 1))
@pytest.mark.parametrize("variant", ["wave", "deep"])
def test_golden_symmetric_cases_windows(name, variant, monkeypatch):
    """The symmetric golden cases of the reference binary as window records, through the wave kernel's window-record instantiation
    and with every tile through pileup_deep_kernel (which reads the same records: raft_testlib.KERNELS)."""
    from raft_amd import engine, hostio
    if variant == "deep":
        monkeypatch.setenv("RAFT_DEEP_MIN", "1")
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = RaftParams(**MAN["synthetic"][name]["params"])
    cols = [z[k] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te")]
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    off = hostio.group_offsets(len(cols[0]), cols[1])
    if off is None:
        pytest.skip(f"{name}: the record stream is not a handful of sorted runs")
    win = hostio.pack_windows(cols[2], cols[3], p.reso)
    if win is None:                                        # (s60_ultralong at -r 10: reads of more than 65,535 windows)
        assert (np.asarray(cols[0], np.int64).max() + p.reso - 1) // p.reso > 65535
        pytest.skip(f"{name}: window indices beyond 16 bits -- the coordinate columns are the caller's form")
    want = oracle_run(p, *cols)
    eng = engine.Engine(sym_params(p), device=0)
    window_runs(eng, p, cols[0], off, win, want, f"{name}/variant {variant}", widths=(4, 1))
    got = result_of(eng, eng.summary)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), (name, k)
    eng.close()


@pytest.mark.parametrize("kw", [dict(n_reads=3000, seed=11), dict(n_reads=6000, seed=12, mean_len=9000.0, coverage=18.0),
                                dict(n_reads=1200, seed=13, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25),
                                dict(n_reads=2500, seed=14, coverage=45.0, copies=5),
                                dict(n_reads=20000, seed=15, mean_len=1500.0, coverage=20.0)])
@pytest.mark.parametrize("tile_bins", [0, 512])
def test_synthetic_sets_windows_vs_oracle(kw, tile_bins):
    """hifiasm-shaped sets (two runs) incl. reads longer than the LDS window (re-cut tiles: pieces clip every record of their
    read), deep repeats (records beyond the prefetched slots) and short reads (tiles of more than 64 reads)."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)))
    want = oracle_run(p, *cols)
    assert want["symmetric"] == 1
    off = hostio.group_offsets(o.n_reads, cols[1])
    win = hostio.pack_windows(cols[2], cols[3], p.reso)
    assert off is not None and off.shape[0] == 2 and win is not None
    eng = engine.Engine(sym_params(p), device=0)
    eng.set_tuning(tile_bins, False)
    window_runs(eng, p, cols[0], off, win, want, f"{kw} tile_bins {tile_bins}")
    eng.close()


@pytest.mark.parametrize("reso", [1, 7, 50, 1000])
@pytest.mark.parametrize("k_runs", [1, 2, 3, 4, 6])
def test_runs_resolutions_and_degenerate_records(k_runs, reso):
    """One to six runs (more than two: unpacked; more than four: merged as well), reads without windows or records,
    intervals without windows (qe == 0, qe <= qs inside one window), window sizes from 1 base up."""
    from raft_amd import engine, hostio
    rng = np.random.default_rng(700 + 10 * k_runs + reso)
    hi = min(40000, 60000 * reso)
    rl = rng.integers(0, hi, 1500).astype(np.int32)
    rl[rng.integers(0, len(rl), 40)] = 0                   # reads without windows
    ok = np.flatnonzero(rl > 0)
    qid = np.concatenate([np.sort(rng.choice(ok, 5000)) for _ in range(k_runs)]).astype(np.int32)
    a = (rng.random(len(qid)) * rl[qid]).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(len(qid)) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
    z = rng.random(len(qid))
    b[z < 0.02] = 0                                        # qe == 0: nothing
    flip = (z >= 0.02) & (z < 0.05)
    b[flip] = a[flip]                                      # qe == qs: window of qs - 1 .. before that of qs: nothing unless they differ
    p = RaftParams(est_cov=8, reso=reso, repeat_length=max(2000, 40 * reso), interval_length=max(2000, 40 * reso), read_length=max(4000, 80 * reso))
    want = oracle_run(p, rl, qid, a, b, qid, a, b); want["symmetric"] = 1
    off = hostio.group_offsets(len(rl), qid, max_runs=16)
    win = hostio.pack_windows(a, b, reso)
    assert off is not None and off.shape[0] == k_runs and win is not None
    w_first, w_last1 = win & 0xffff, win >> 16
    live = (b > 0) & ((b - 1) // reso + 1 > a // reso)
    assert np.array_equal(win == 0, ~live) and np.array_equal(w_first[live], (a // reso)[live]) and np.array_equal(w_last1[live], ((b - 1) // reso + 1)[live])
    eng = engine.Engine(sym_params(p), device=0)
    window_runs(eng, p, rl, off, win, want, f"{k_runs} runs, reso {reso}", widths=(4, 1))
    if k_runs <= 4:
        res, s = eng.run_pipelined_windows(rl, off, win, n_chunks=3)
        check_pipelined(res, s, want, f"{k_runs} runs, reso {reso}: pipelined")
    eng.close()


def test_errors_of_window_records():
    import torch
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=3000, seed=21)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    off = hostio.group_offsets(o.n_reads, cols[1])
    win = hostio.pack_windows(cols[2], cols[3], p.reso)
    eng = engine.Engine(sym_params(p), device=0)
    t = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to("cuda:0")
    d_rl, d_off = t(cols[0]), t(off, torch.int64)
    B = n_windows(p, cols[0])
    # (a) a record reaching past the last window of its read: same code and record index as from the coordinate entries
    be = cols[3].copy(); be[4321] = cols[0][cols[1][4321]] + 7000
    wbad = hostio.pack_windows(cols[2], be, p.reso)
    from raft_testlib import kernel_mode
    for variant in ("wave", "deep"):
        with kernel_mode(variant):
            for hint in (B, -1):
                with pytest.raises(engine.RaftError) as e1:
                    eng.run_device_windows(d_rl, d_off, t(wbad.view(np.int32)), n_bins=hint); eng.finish()
                assert e1.value.code == engine.ERR_COORD and e1.value.index == 4321, (variant, hint)
    with pytest.raises(engine.RaftError) as e2:
        eng.run_host(cols[0], cols[1], cols[2], be, None, None, None); eng.finish()
    assert e2.value.code == engine.ERR_COORD and e2.value.index == 4321
    with pytest.raises(engine.RaftError) as e3:
        eng.run_pipelined_windows(cols[0], off, wbad, n_chunks=5)
    assert e3.value.code == engine.ERR_COORD and e3.value.index == 4321
    # (b) what window records cannot say is the tokeniser's to report
    neg = cols[2].copy(); neg[17] = -3; neg[900] = -1
    with pytest.raises(hostio.HostError) as h:
        hostio.pack_windows(neg, cols[3], p.reso)
    assert h.value.code == hostio.ERR_COORD and h.value.index == 17
    far = cols[3].copy(); far[5] = 65536 * p.reso
    assert hostio.pack_windows(cols[2], far, p.reso) is None
    far[5] = 65535 * p.reso
    assert hostio.pack_windows(cols[2], far, p.reso) is not None
    # (c) offsets that step back / do not chain; a negative read length; a context without the symmetric flag
    bad = off.copy(); bad[0, 100] = bad[0, 101] + 5
    for hint in (B, -1):
        with pytest.raises(engine.RaftError) as e:
            eng.run_device_windows(d_rl, t(bad, torch.int64), t(win.view(np.int32)), n_bins=hint); eng.finish()
        assert e.value.code == engine.ERR_PARAM
    neg_len = cols[0].copy(); neg_len[77] = -5
    for hint in (B, -1):
        with pytest.raises(engine.RaftError) as e4:
            eng.run_device_windows(t(neg_len), d_off, t(win.view(np.int32)), n_bins=hint); eng.finish()
        assert e4.value.code == engine.ERR_PARAM and e4.value.index == 77
    e5 = engine.Engine(p, device=0)
    with pytest.raises(engine.RaftError) as e:
        e5.run_device_windows(d_rl, d_off, t(win.view(np.int32)))
    assert e.value.code == engine.ERR_PARAM
    e5.close()
    e6 = engine.Engine(RaftParams(est_cov=30, reso=40000, repeat_length=80000, interval_length=80000, read_length=160000, symmetric_mode=1), device=0)
    with pytest.raises(engine.RaftError) as e:
        e6.run_device_windows(d_rl, d_off, t(win.view(np.int32)))
    assert e.value.code == engine.ERR_PARAM                # (reso above 32767: the coordinate columns)
    e6.close()
    # the context is still good
    eng.run_device_windows(d_rl, d_off, t(win.view(np.int32)), n_bins=B)
    assert_same_result(result_of(eng, eng.finish()), want, "after the errors")
    # no records at all
    eng.run_device_windows(d_rl, t(np.zeros((1, o.n_reads + 1), np.int64), torch.int64), t(np.zeros(0, np.int32)), n_bins=B)
    s = eng.finish()
    assert s.total_coverage == 0 and s.n_repeats == 0 and s.n_bins == B
    eng.close()


def test_window_records_when_tiles_have_no_slots_of_their_own(monkeypatch):
    """RAFT_EXTRA_CAP=1 (rounds 3-5: the re-cut list overflowed and the general kernel took the pass; now: tile ids without slots of
    their own in a four-bit pass) on a long-read set as window records: the oracle's arrays."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=1200, seed=13, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=25)
    want = oracle_run(p, *cols)
    off = hostio.group_offsets(o.n_reads, cols[1])
    win = hostio.pack_windows(cols[2], cols[3], p.reso)
    monkeypatch.setenv("RAFT_EXTRA_CAP", "1")
    eng = engine.Engine(sym_params(p), device=0)
    window_runs(eng, p, cols[0], off, win, want, "extra-tile overflow", widths=(4, 1), with_host=False)
    eng.close()


def check_pipelined(res, s, want, what):
    from raft_amd import hostio
    assert np.array_equal(hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"]), want["cov"]), what
    for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
        assert np.array_equal(res[k], want[k]), (what, k)
    assert (s.symmetric, s.high_cov, s.total_coverage, s.total_windows, s.total_repeat_length, s.total_read_length) == \
        tuple(want[k] for k in ("symmetric", "high_cov", "total_coverage", "total_windows", "total_repeat_length", "total_read_length")), what
    assert s.n_fragments == len(want["frag_read"]) and s.n_repeats == len(want["rep_s"]) and s.n_intervals == want["n_intervals"], what


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
@pytest.mark.parametrize("kw,n_chunks", [(dict(n_reads=4000, seed=81), 5), (dict(n_reads=4000, seed=81), 2), (dict(n_reads=4000, seed=81), 23),
                                         (dict(n_reads=1500, seed=82, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25), 4),
                                         (dict(n_reads=50000, seed=2), 0)])
def test_pipelined_windows_equals_oracle(kw, n_chunks, n_ctx):
    """The chunked host pipeline / several contexts on window records: one column and the offsets' slices go up."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)))
    want = oracle_run(p, *cols)
    off = hostio.group_offsets(o.n_reads, cols[1])
    win = hostio.pack_windows(cols[2], cols[3], p.reso)
    eng = engine.Engine(sym_params(p), device=0)
    others = [engine.Engine(RaftParams(est_cov=3, reso=7, symmetric_mode=1), device=0) for _ in range(n_ctx - 1)]
    out = eng.host_output_buffers(cols[0], pinned=True)
    for rep in range(2):
        res, s = eng.run_pipelined_windows(cols[0], off, win, n_chunks=n_chunks, out=out, others=others)
        check_pipelined(res, s, want, f"{kw} chunks {n_chunks} contexts {n_ctx} pass {rep}")
        assert s.n_segments == 2
    for e2 in [eng] + others:
        e2.close()


def test_windows_equal_plain_on_the_bench_workload_slice():
    """A 412 k-read slice of BASELINE configs[2]: the window-record pass against the plain detecting pass on the device, array
    by array, in every output width."""
    import torch
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(412_500, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
    p = RaftParams(est_cov=32)
    e0 = engine.Engine(p, device=0)
    e0.run_device(o.read_len, *o.columns())
    s0 = e0.finish()
    a = {k: v.clone() for k, v in e0.outputs_device().items()}
    off = hostio.group_offsets(o.n_reads, o.qid.cpu().numpy())
    win = hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), p.reso)
    assert off is not None and off.shape[0] == 2 and win is not None
    B = n_windows(p, o.read_len.cpu().numpy())
    e1 = engine.Engine(sym_params(p), device=0)
    d_off = torch.as_tensor(off).to("cuda:0")
    d_w = torch.as_tensor(win.view(np.int32)).to("cuda:0")
    for width in (4, 1, 2):
        e1.set_output_width(width)
        e1.run_device_windows(o.read_len, d_off, d_w, n_bins=B)
        s1 = e1.finish()
        b = e1.outputs_device()
        for k in a:
            assert torch.equal(a[k], b[k]), (k, width)
        for f in ("n_bins", "n_repeats", "n_cuts", "n_fragments", "total_coverage", "total_repeat_length", "total_read_length", "n_intervals"):
            assert getattr(s0, f) == getattr(s1, f), f
    e0.close(); e1.close()
