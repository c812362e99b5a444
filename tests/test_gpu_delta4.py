"""GPU: the four-bit step encoding of the coverage array (include/raft_hip.h "delta4") -- written by the pileup kernel itself
(raft_hip_set_output_width(8)), made from an int32 pass on request (raft_hip_fetch_delta4), decoded on the device for
raft_hip_fetch / outputs_device and on the host by raft_host_unpack_coverage_d4 -- against the CPU oracle.

cov[w] - cov[w-1] is the pileup's own difference array; windows whose step leaves [-7, 7] and every tile's first window are
listed with their value; a decoder starts at any multiple of 1024 windows.  Bar: bit-exact.
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


def result_of(eng, s):
    got = eng.fetch()
    got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
               total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    return got


def check_encoding(d4, want, what):
    """The encoding as fetched: decodes to the oracle's array; exceptions ascending, each on an escaped window with its value;
    anchors are the values before the blocks; steps within [-7, 7] are never listed except at escapes the kernel forces."""
    from raft_amd import hostio
    cov = want["cov"]
    n = cov.size
    got = hostio.unpack_coverage_d4(n, d4["cov_nib"], d4["cov_anchor"], d4["exc_index"], d4["exc_value"])
    assert np.array_equal(got, cov), what
    xi = d4["exc_index"]
    assert np.all(np.diff(xi) > 0) and np.array_equal(d4["exc_value"], cov[xi]), what
    if n:
        codes = np.stack([d4["cov_nib"] & 15, d4["cov_nib"] >> 4], 1).reshape(-1)[:n]
        assert np.array_equal(np.flatnonzero(codes == 0), xi), what
        step = np.diff(np.concatenate([[0], cov.astype(np.int64)]))
        assert np.array_equal(codes[codes != 0].astype(np.int64) - 8, step[codes != 0]), what
        assert np.all(np.abs(step[codes != 0]) <= 7) and np.all(codes[np.abs(step) > 7] == 0), what
        # anchors: the value before the block -- needed (and checked) where the block's first window is a step, not a listed value
        an, first = d4["cov_anchor"], codes[0::1024]
        true = np.concatenate([[0], cov[1023::1024]])[: an.size]
        assert an.size == (n + 1023) // 1024 and np.array_equal(an[first != 0], true[first != 0]), what
    for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_read", "frag_begin", "frag_end"):
        assert np.array_equal(d4[k], want[k]), (what, k)


def all_forms(eng, run, want, what, direct=True):
    """An int32 pass encoded afterwards, and passes that write the encoding themselves; device decode and host decode."""
    eng.set_output_width(4)
    s = run()
    check_encoding(eng.fetch_delta4(), want, what + ": encoded after an int32 pass")
    assert_same_result(result_of(eng, s), want, what + ": int32 after the encoding was made")
    for w in (8, 1, 8):
        eng.set_output_width(w)
        s = run()
        pk = eng.packed_device()
        if direct and w == 8:
            assert pk is not None and pk["width"] == 8, what
        check_encoding(eng.fetch_delta4(), want, f"{what}: pass in width {w}")
        assert_same_result(result_of(eng, s), want, f"{what}: decoded on the device after a pass in width {w}")
    eng.set_output_width(4)


@pytest.mark.parametrize("name", sorted(MAN["synthetic"]))
@pytest.mark.parametrize("variant", ["wave", "deep"])
def test_golden_cases_delta4(name, variant, monkeypatch):
    from raft_amd import engine
    if variant == "deep":
        monkeypatch.setenv("RAFT_DEEP_MIN", "1")          # (every tile through pileup_deep_kernel: raft_testlib.KERNELS)
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = RaftParams(**MAN["synthetic"][name]["params"])
    cols = [z[k] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te")]
    want = oracle_run(p, *cols)
    eng = engine.Engine(p, device=0)

    def run():
        eng.run_host(*cols)
        return eng.finish()
    all_forms(eng, run, want, f"{name}/variant {variant}", direct=False)
    eng.close()


@pytest.mark.parametrize("kw", [dict(n_reads=3000, seed=11), dict(n_reads=6000, seed=12, mean_len=9000.0, coverage=18.0),
                                dict(n_reads=1200, seed=13, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25),
                                dict(n_reads=2500, seed=14, coverage=45.0, copies=5),
                                dict(n_reads=20000, seed=15, mean_len=1500.0, coverage=20.0),
                                dict(n_reads=1500, seed=16, mean_len=12000.0, coverage=400.0, n_families=0)])
@pytest.mark.parametrize("tile_bins", [0, 512, 1000])
@pytest.mark.parametrize("form", ["columns", "grouped", "windows"])
def test_synthetic_sets_delta4(kw, tile_bins, form):
    """Every input form of the default configuration writes the encoding directly: tiles whose first window shares a ushort
    with the previous tile's last (tile quanta that are no multiple of four), long reads in pieces, deep sets where most steps
    at read boundaries escape."""
    import torch
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)))
    want = oracle_run(p, *cols)
    eng = engine.Engine(sym_params(p), device=0)
    eng.set_tuning(tile_bins, False)
    dev = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to("cuda:0")
    d_rl = dev(cols[0])
    if form == "columns":
        d = [dev(c) for c in cols[1:4]]
        def run():
            eng.run_device(d_rl, d[0], d[1], d[2], None, None, None)
            return eng.finish()
    else:
        off = hostio.group_offsets(o.n_reads, cols[1])
        d_off = dev(off, torch.int64)
        B = int(((cols[0].astype(np.int64) + p.reso - 1) // p.reso).sum())
        if form == "grouped":
            d = [dev(c) for c in cols[2:4]]
            def run():
                eng.run_device_grouped(d_rl, d_off, None, d[0], d[1], n_bins=B)
                return eng.finish()
        else:
            d_w = dev(hostio.pack_windows(cols[2], cols[3], p.reso).view(np.int32))
            def run():
                eng.run_device_windows(d_rl, d_off, d_w, n_bins=B)
                return eng.finish()
    all_forms(eng, run, want, f"{kw} tile_bins {tile_bins} {form}")
    eng.close()


def test_exception_list_overflow_and_empty_inputs():
    """More escaped windows than the first list holds (one-window reads of alternating depth: every window is a large step):
    the pass is run again with room.  A 400-deep set: steps stay small however deep the pile -- a handful of escapes.  No reads,
    reads without windows, no records."""
    import torch
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    n = 30000
    rl = np.full(n, 40, np.int32)
    depth = np.where(np.arange(n) % 2 == 0, 3, 25)
    qid = np.repeat(np.arange(n, dtype=np.int32), depth)
    qs, qe = np.zeros(qid.size, np.int32), np.full(qid.size, 40, np.int32)
    p = RaftParams(est_cov=10)
    want = oracle_run(p, rl, qid, qs, qe, qid, qs, qe); want["symmetric"] = 1
    eng = engine.Engine(sym_params(p), device=0)
    eng.set_output_width(8)
    eng.run_host(rl, qid, qs, qe, None, None, None)
    s = eng.finish()
    d4 = eng.fetch_delta4()
    check_encoding(d4, want, "one-window reads")
    assert d4["exc_index"].size == n > max(4096, n // 64)            # (more than the first list's room: the pass ran twice)
    assert_same_result(result_of(eng, s), want, "one-window reads, decoded on the device")
    o = make_overlaps(n_reads=1500, seed=16, mean_len=12000.0, coverage=400.0, n_families=0)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=400)
    want = oracle_run(p, *cols)
    eng.close()
    eng = engine.Engine(sym_params(p), device=0)
    eng.set_output_width(8)
    eng.run_host(cols[0], cols[1], cols[2], cols[3], None, None, None)
    s = eng.finish()
    d4 = eng.fetch_delta4()
    check_encoding(d4, want, "deep set")
    assert d4["exc_index"].size < want["cov"].size // 100 and int(want["cov"].max()) > 300
    # degenerate inputs
    for rl in (np.zeros(0, np.int32), np.zeros(5, np.int32), np.array([49, 50, 51, 1, 0, 2048 * 50], np.int32)):
        none = np.zeros(0, np.int32)
        w = oracle_run(p, rl, none, none, none, none, none, none)
        eng.run_host(rl, none, none, none, None, None, None)
        s = eng.finish()
        check_encoding(eng.fetch_delta4(), w, f"no records, reads {rl.tolist()}")
        assert_same_result(result_of(eng, s), dict(w, symmetric=1), "no records")
    eng.close()


def test_delta4_on_the_bench_workload_slice():
    """A 412 k-read slice of BASELINE configs[2]: the pass writing the encoding against the int32 pass on the device."""
    import torch
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(412_500, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
    p = RaftParams(est_cov=32, symmetric_mode=1)
    e0 = engine.Engine(p, device=0)
    e0.run_device(o.read_len, o.qid, o.qs, o.qe, None, None, None)
    s0 = e0.finish()
    a = {k: v.clone() for k, v in e0.outputs_device().items()}
    off = torch.as_tensor(hostio.group_offsets(o.n_reads, o.qid.cpu().numpy())).to("cuda:0")
    win = torch.as_tensor(hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), p.reso).view(np.int32)).to("cuda:0")
    e1 = engine.Engine(p, device=0)
    e1.set_output_width(8)
    for form in ("columns", "windows"):
        if form == "columns":
            e1.run_device(o.read_len, o.qid, o.qs, o.qe, None, None, None)
        else:
            e1.run_device_windows(o.read_len, off, win, n_bins=s0.n_bins)
        s1 = e1.finish()
        pk = e1.packed_device()
        assert pk["width"] == 8
        n_exc = int(pk["exc_index"].numel())
        assert n_exc < s0.n_bins // 100, (form, n_exc)            # (0.2-0.3 % of the windows on this set)
        b = e1.outputs_device()
        for k in a:
            assert torch.equal(a[k], b[k]), (k, form)
        for f in ("n_bins", "n_repeats", "n_cuts", "n_fragments", "total_coverage", "total_repeat_length", "total_read_length"):
            assert getattr(s0, f) == getattr(s1, f), f
    e0.close(); e1.close()


def check_pipelined_d4(res, s, want, what):
    from raft_amd import hostio
    n = want["cov"].size
    got = hostio.unpack_coverage_d4(n, res["cov_nib"], res["cov_anchor"], res["exc_index"], res["exc_value"])
    assert np.array_equal(got, want["cov"]), what
    assert np.all(np.diff(res["exc_index"]) > 0) and np.array_equal(res["exc_value"], want["cov"][res["exc_index"]]), what
    for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
        assert np.array_equal(res[k], want[k]), (what, k)
    assert (s.symmetric, s.high_cov, s.total_coverage, s.total_windows, s.total_repeat_length, s.total_read_length) == \
        tuple(want[k] for k in ("symmetric", "high_cov", "total_coverage", "total_windows", "total_repeat_length", "total_read_length")), what
    assert s.n_fragments == len(want["frag_read"]) and s.n_repeats == len(want["rep_s"]), what


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
@pytest.mark.parametrize("kw,n_chunks", [(dict(n_reads=4000, seed=81), 5), (dict(n_reads=4000, seed=81), 2), (dict(n_reads=9000, seed=83, mean_len=6000.0), 23),
                                         (dict(n_reads=1500, seed=82, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25), 4),
                                         (dict(n_reads=50000, seed=2), 0), (dict(n_reads=300, seed=84), 3)])
@pytest.mark.parametrize("form", ["columns", "grouped", "windows"])
def test_pipelined_delta4_equals_oracle(kw, n_chunks, n_ctx, form):
    """The host pipelines handing the coverage back as four-bit steps: chunk boundaries move to reads that begin on a multiple
    of 1024 windows (whole anchor blocks, whole bytes), several contexts write disjoint ranges of the caller's arrays; a set
    too small to be cut goes through in one piece."""
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
    out = eng.host_output_buffers(cols[0], pinned=True, width=8)
    for rep in range(2):
        out["cov_nib"][:] = 0xA5; out["cov_anchor"][:] = -7           # (stale bytes of the pass before must not survive)
        if form == "columns":
            res, s = eng.run_pipelined(cols[0], cols[1], cols[2], cols[3], n_chunks=n_chunks, out=out, others=others)
        elif form == "grouped":
            res, s = eng.run_pipelined_grouped(cols[0], off, cols[2], cols[3], n_chunks=n_chunks, out=out, others=others)
        else:
            res, s = eng.run_pipelined_windows(cols[0], off, win, n_chunks=n_chunks, out=out, others=others)
        check_pipelined_d4(res, s, want, f"{kw} chunks {n_chunks} contexts {n_ctx} {form} pass {rep}")
    for e2 in [eng] + others:
        e2.close()


def test_pipelined_delta4_errors_and_shuffled_input():
    """A data error comes back as from the byte encodings; a shuffled stream (no chunk plan; with several contexts the
    host-routed path, which cannot align its chunks) is served in one piece."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=4000, seed=81)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    off = hostio.group_offsets(o.n_reads, cols[1])
    eng = engine.Engine(sym_params(p), device=0)
    other = engine.Engine(sym_params(p), device=0)
    out = eng.host_output_buffers(cols[0], pinned=False, width=8)
    be = cols[3].copy(); be[4321] = cols[0][cols[1][4321]] + 7000
    with pytest.raises(engine.RaftError) as e:
        eng.run_pipelined_grouped(cols[0], off, cols[2], be, n_chunks=5, out=out)
    assert e.value.code == engine.ERR_COORD and e.value.index == 4321
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(cols[1])); perm = np.concatenate([[0], perm[perm != 0]])
    sh = [c[perm] for c in cols[1:4]]
    res, s = eng.run_pipelined(cols[0], sh[0], sh[1], sh[2], n_chunks=4, out=out, others=[other])
    check_pipelined_d4(res, s, want, "shuffled stream, two contexts")
    small = {k: v[: max(1, v.size // 8)] for k, v in out.items()}
    with pytest.raises(engine.RaftError) as e:
        eng.run_pipelined_grouped(cols[0], off, cols[2], cols[3], n_chunks=5, out=small)
    assert e.value.code == engine.ERR_TOO_LARGE
    eng.close(); other.close()


def test_delta4_when_tiles_have_no_slots_of_their_own(monkeypatch):
    """RAFT_EXTRA_CAP=1: one tile id has slots of its own for the windows it lists, every other tile lists in the shared list (rounds
    3-5: the re-cut list overflowed and the general kernel took the pass) -- the same encoding."""
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    o = make_overlaps(n_reads=1200, seed=13, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=25)
    want = oracle_run(p, *cols)
    monkeypatch.setenv("RAFT_EXTRA_CAP", "1")
    eng = engine.Engine(sym_params(p), device=0)
    eng.set_output_width(8)
    eng.run_host(cols[0], cols[1], cols[2], cols[3], None, None, None)
    s = eng.finish()
    check_encoding(eng.fetch_delta4(), want, "extra-tile overflow")
    assert_same_result(result_of(eng, s), want, "extra-tile overflow, int32")
    res, s2 = eng.run_pipelined(cols[0], cols[1], cols[2], cols[3], n_chunks=3, out=eng.host_output_buffers(cols[0], pinned=False, width=8))
    check_pipelined_d4(res, s2, want, "extra-tile overflow, pipelined")
    eng.close()
