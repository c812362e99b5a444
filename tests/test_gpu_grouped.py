"""GPU: the grouped-input form of the boundary (raft_hip_run_device_grouped / run_host_grouped / run_multi_grouped) against
the CPU oracle, the golden outputs of the reference binary and the plain six-column entry.

Grouped input = what a tokeniser that resolves every name knows anyway (reference README.md:36-38: hifiasm writes its PAF
grouped by query; chop.hpp:147-169 meets the records in that order): per sorted run, where every read's records begin.
The pass then needs no look at the stream, no searches for its tile cuts, no query column over PCIe, and -- with the
caller's window count -- no host wait.  Bar: bit-exact.
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


def grouped_runs(eng, p, rl, qid, qs, qe, off, want, what, with_host=True):
    """Every form of the grouped entry on one context; each must equal `want`."""
    import torch
    dev = "cuda:0"
    t = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(dev)
    d_rl, d_q, d_s, d_e, d_off = t(rl), t(qid), t(qs), t(qe), t(off, torch.int64)
    B = n_windows(p, rl)
    for form, kw in (("qid + hint", dict(qid=d_q, n_bins=B)), ("qid, no hint", dict(qid=d_q, n_bins=-1)),
                     ("no qid + hint", dict(qid=None, n_bins=B)), ("no qid, no hint", dict(qid=None, n_bins=-1)),
                     ("wrong hint (low)", dict(qid=d_q, n_bins=max(B - 17, 0))), ("wrong hint (high)", dict(qid=None, n_bins=B + 4096))):
        eng.run_device_grouped(d_rl, d_off, kw["qid"], d_s, d_e, n_bins=kw["n_bins"])
        s = eng.finish()
        assert_same_result(result_of(eng, s), want, f"{what}: {form}")
        assert s.interval_path == 0 and s.n_segments == off.shape[0] and s.n_bins == B
    if with_host:
        eng.run_host_grouped(rl, off, qs, qe)
        s = eng.finish()
        assert_same_result(result_of(eng, s), want, f"{what}: run_host_grouped")


@pytest.mark.parametrize("name", sorted(n for n, m in MAN["synthetic"].items() if m["symmetric"] == 1))
@pytest.mark.parametrize("variant", ["wave", "deep"])
def test_golden_symmetric_cases_grouped(name, variant, monkeypatch):
    """The symmetric golden cases of the reference binary through the grouped entry (when their record stream is a handful
    of sorted runs; a shuffled one has no grouped form and group_offsets says so)."""
    from raft_amd import engine, hostio
    if variant == "deep":
        monkeypatch.setenv("RAFT_DEEP_MIN", "1")          # (every tile through pileup_deep_kernel: raft_testlib.KERNELS)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = RaftParams(**MAN["synthetic"][name]["params"])
    cols = [z[k] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te")]
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    off = hostio.group_offsets(len(cols[0]), cols[1])
    if off is None:
        pytest.skip(f"{name}: the record stream is not a handful of sorted runs")
    want = oracle_run(p, *cols)
    eng = engine.Engine(sym_params(p), device=0)
    grouped_runs(eng, p, cols[0], cols[1], cols[2], cols[3], off, want, f"{name}/variant {variant}")
    got = result_of(eng, eng.summary)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), (name, k)
    eng.close()


@pytest.mark.parametrize("kw", [dict(n_reads=3000, seed=11), dict(n_reads=6000, seed=12, mean_len=9000.0, coverage=18.0),
                                dict(n_reads=1200, seed=13, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25),
                                dict(n_reads=2500, seed=14, coverage=45.0, copies=5)])
@pytest.mark.parametrize("tile_bins", [0, 512])
def test_synthetic_sets_grouped_vs_oracle(kw, tile_bins):
    """hifiasm-shaped sets (cis + trans file, each grouped by query): two runs; incl. reads longer than the LDS window
    (re-cut tiles use the offsets for their sub-ranges) and deep repeats."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)))
    want = oracle_run(p, *cols)
    assert want["symmetric"] == 1
    off = hostio.group_offsets(o.n_reads, cols[1])
    assert off is not None and off.shape == (2, o.n_reads + 1) and off[0, 0] == 0 and off[1, -1] == o.n_rec and off[0, -1] == off[1, 0] == o.n_cis
    eng = engine.Engine(sym_params(p), device=0)
    eng.set_tuning(tile_bins, False)
    grouped_runs(eng, p, cols[0], cols[1], cols[2], cols[3], off, want, f"{kw} tile_bins {tile_bins}")
    eng.close()


@pytest.mark.parametrize("k_runs", [1, 3, 4])
def test_one_to_four_runs_and_reads_without_records(k_runs):
    from raft_amd import engine, hostio
    rng = np.random.default_rng(500 + k_runs)
    rl = rng.integers(0, 40000, 1500).astype(np.int32)
    rl[rng.integers(0, len(rl), 40)] = 0                   # reads without windows
    ok = np.flatnonzero(rl > 0)
    qid = np.concatenate([np.sort(rng.choice(ok, 6000)) for _ in range(k_runs)]).astype(np.int32)
    a = (rng.random(len(qid)) * rl[qid]).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(len(qid)) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
    p = RaftParams(est_cov=8)
    want = oracle_run(p, rl, qid, a, b, qid, a, b); want["symmetric"] = 1
    off = hostio.group_offsets(len(rl), qid)
    assert off is not None and off.shape[0] == k_runs
    eng = engine.Engine(sym_params(p), device=0)
    grouped_runs(eng, p, rl, qid, a, b, off, want, f"{k_runs} runs")
    eng.close()
    # five runs: not a grouped stream for this engine
    q5 = np.concatenate([np.sort(rng.choice(ok, 100)) for _ in range(5)]).astype(np.int32)
    assert hostio.group_offsets(len(rl), q5) is None


def test_offsets_that_disagree_with_the_query_column_and_bad_offsets():
    """With a query column at hand every record is checked against its tile's reads: offsets that do not describe the stream
    send the pass to the plain form (same results); offsets that step back or do not chain are the caller's error."""
    import torch
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=3000, seed=21)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    off = hostio.group_offsets(o.n_reads, cols[1])
    eng = engine.Engine(sym_params(p), device=0)
    t = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to("cuda:0")
    d = [t(cols[0]), t(cols[1]), t(cols[2]), t(cols[3])]
    B = n_windows(p, cols[0])
    # (a) monotone, chained, but wrong: every inner offset of run 0 moved by a few hundred records
    wrong = off.copy()
    wrong[0, 1:-1] = np.minimum(wrong[0, 1:-1] + 300, wrong[0, -1])
    wrong[1, 1:-1] = np.maximum(wrong[1, 1:-1] - 200, wrong[1, 0])
    for hint in (B, -1):
        eng.run_device_grouped(d[0], t(wrong, torch.int64), d[1], d[2], d[3], n_bins=hint)
        s = eng.finish()
        assert_same_result(result_of(eng, s), want, f"wrong offsets, hint {hint}")
    # (b) offsets that step back / do not start at 0 / do not end at n_rec
    for what, mut in (("step back", lambda x: x.__setitem__((0, 100), x[0, 101] + 5)),
                      ("start", lambda x: x.__setitem__((0, 0), 1)),
                      ("end", lambda x: x.__setitem__((1, -1), x[1, -1] - 1)),
                      ("chain", lambda x: x.__setitem__((1, 0), x[1, 0] + 1))):
        bad = off.copy()
        mut(bad)
        for hint in (B, -1):
            for q in (d[1], None):
                with pytest.raises(engine.RaftError) as e:
                    eng.run_device_grouped(d[0], t(bad, torch.int64), q, d[2], d[3], n_bins=hint)
                    eng.finish()
                assert e.value.code == engine.ERR_PARAM, (what, hint, e.value.code)
        with pytest.raises(engine.RaftError) as e:
            eng.run_pipelined_grouped(cols[0], bad, cols[2], cols[3], n_chunks=4)
        assert e.value.code == engine.ERR_PARAM, what
    # (c) data errors come back as from the plain entry: same code, same record index
    be = cols[3].copy(); be[4321] = cols[0][cols[1][4321]] + 7000
    with pytest.raises(engine.RaftError) as e1:
        eng.run_device_grouped(d[0], t(off, torch.int64), None, d[2], t(be), n_bins=B); eng.finish()
    with pytest.raises(engine.RaftError) as e2:
        eng.run_host(cols[0], cols[1], cols[2], be, None, None, None); eng.finish()
    assert e1.value.code == e2.value.code == engine.ERR_COORD and e1.value.index == e2.value.index == 4321
    with pytest.raises(engine.RaftError) as e3:
        eng.run_pipelined_grouped(cols[0], off, cols[2], be, n_chunks=5)
    assert e3.value.code == engine.ERR_COORD and e3.value.index == 4321
    neg = cols[0].copy(); neg[77] = -5
    for hint in (B, -1):
        with pytest.raises(engine.RaftError) as e4:
            eng.run_device_grouped(t(neg), t(off, torch.int64), d[1], d[2], d[3], n_bins=hint); eng.finish()
        assert e4.value.code == engine.ERR_PARAM and e4.value.index == 77
    # the context is still good
    eng.run_device_grouped(d[0], t(off, torch.int64), None, d[2], d[3], n_bins=B)
    assert_same_result(result_of(eng, eng.finish()), want, "after the errors")
    # a context that does not assert the symmetric flag cannot take grouped input
    e5 = engine.Engine(p, device=0)
    with pytest.raises(engine.RaftError) as e:
        e5.run_device_grouped(d[0], t(off, torch.int64), d[1], d[2], d[3])
    assert e.value.code == engine.ERR_PARAM
    e5.close(); eng.close()


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
def test_pipelined_grouped_equals_oracle(kw, n_chunks, n_ctx):
    """The chunked host pipeline / several contexts on grouped input: offsets' slices instead of the query column."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)))
    want = oracle_run(p, *cols)
    off = hostio.group_offsets(o.n_reads, cols[1])
    eng = engine.Engine(sym_params(p), device=0)
    others = [engine.Engine(RaftParams(est_cov=3, reso=7, symmetric_mode=1), device=0) for _ in range(n_ctx - 1)]
    out = eng.host_output_buffers(cols[0], pinned=True)
    for rep in range(2):
        res, s = eng.run_pipelined_grouped(cols[0], off, cols[2], cols[3], n_chunks=n_chunks, out=out, others=others)
        check_pipelined(res, s, want, f"{kw} chunks {n_chunks} contexts {n_ctx} pass {rep}")
        assert s.n_segments == 2
    for e2 in [eng] + others:
        e2.close()


def test_grouped_equals_plain_on_the_bench_workload_slice():
    """A 412 k-read slice of BASELINE configs[2] (what one of eight GPUs holds in configs[3]): the grouped, hint-sized pass
    against the plain detecting pass on the device, array by array; exceptions spread over several contexts."""
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
    assert off is not None and off.shape[0] == 2
    B = n_windows(p, o.read_len.cpu().numpy())
    e1 = engine.Engine(sym_params(p), device=0)
    d_off = torch.as_tensor(off).to("cuda:0")
    for q in (o.qid, None):
        e1.run_device_grouped(o.read_len, d_off, q, o.qs, o.qe, n_bins=B)
        s1 = e1.finish()
        b = e1.outputs_device()
        for k in a:
            assert torch.equal(a[k], b[k]), (k, q is None)
        for f in ("n_bins", "n_repeats", "n_cuts", "n_fragments", "total_coverage", "total_repeat_length", "total_read_length", "n_intervals"):
            assert getattr(s0, f) == getattr(s1, f), f
    e0.close(); e1.close()
