"""GPU: the HIP engine (through the C ABI) against the CPU oracle and the golden reference outputs.

Bar: bit-exact (integer / index path, no tolerance).
"""
import json
import os

import numpy as np
import pytest
from raft_testlib import GOLDEN, OracleError, RaftParams, assert_same_result, oracle_run

pytestmark = pytest.mark.gpu

MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))


@pytest.fixture(scope="module")
def eng_mod():
    from raft_amd import engine
    return engine


def run_engine(engine, p, cols, tile_bins=0, force_bucket=False, device_resident=False, variant="wave"):
    """variant: "wave" (the pass as it comes) or "deep" (every tile through pileup_deep_kernel), raft_testlib.KERNELS."""
    from raft_testlib import kernel_mode
    eng = engine.Engine(p, device=0)
    try:
        eng.set_tuning(tile_bins, force_bucket, -1)
        with kernel_mode(variant):
            if device_resident:
                import torch
                dev = [torch.as_tensor(np.ascontiguousarray(c, dtype=np.int32)).to("cuda:0") for c in cols]
                eng.use_torch_stream()
                eng.run_device(*dev)
            else:
                eng.run_host(*cols)
            s = eng.finish()
        got = eng.fetch()
        got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage,
                   total_windows=s.total_windows, total_repeat_length=s.total_repeat_length,
                   total_read_length=s.total_read_length)
        return got, s
    finally:
        eng.close()


def test_wave_primitives_selftest(eng_mod):
    assert eng_mod.selftest(0) == 0


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = MAN["synthetic"][name]
    p = RaftParams(**meta["params"])
    cols = [z[k] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te")]
    exp = {k[4:]: z[k] for k in z.files if k.startswith("exp_")}
    return p, cols, exp, meta


@pytest.mark.parametrize("name", sorted(MAN["synthetic"]))
@pytest.mark.parametrize("mode", ["auto", "bucket", "tile64", "device"])
def test_golden_cases(eng_mod, name, mode):
    p, cols, exp, meta = load_case(name)
    want = oracle_run(p, *cols)
    got, s = run_engine(eng_mod, p, cols, tile_bins=64 if mode == "tile64" else 0, force_bucket=(mode == "bucket"),
                        device_resident=(mode == "device"))
    assert_same_result(got, want, f"{name}/{mode} vs oracle")
    for k in exp:                       # and directly against what the compiled reference wrote
        assert np.array_equal(got[k], exp[k]), (name, mode, k)
    assert s.symmetric == meta["symmetric"]
    assert s.n_records == meta["n_rec"]
    if mode == "bucket":
        assert s.interval_path == 1


def random_case(seed, n_lo=1, n_hi=60, len_hi=3000, m_hi=600):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(n_lo, n_hi))
    reso = int(rng.choice([1, 3, 7, 50, 64]))
    rl = rng.integers(0, len_hi, n).astype(np.int32)
    if seed % 5 == 0:
        rl[rng.integers(0, n)] = int(rng.integers(20000, 90000))   # one long read -> chunk mode at small reso
    m = int(rng.integers(1, m_hi))
    qid = rng.integers(0, n, m).astype(np.int32)
    tid = rng.integers(0, n, m).astype(np.int32)
    if seed % 3 == 0:
        qid.sort()

    def coords(ids):
        ln = rl[ids].astype(np.int64)
        hi = ((ln + reso - 1) // reso) * reso
        a = (rng.random(m) * (hi + 1)).astype(np.int64)
        b = (rng.random(m) * (hi + 1)).astype(np.int64)
        s, e = np.minimum(a, b), np.maximum(a, b)
        inv = rng.random(m) < 0.1
        return np.where(inv, e, s).astype(np.int32), np.where(inv, s, e).astype(np.int32)

    qs, qe = coords(qid)
    ts, te = coords(tid)
    if seed % 2 == 0 and m > 3:
        k = int(rng.integers(1, m))
        qid[k], tid[k], qs[k], qe[k], ts[k], te[k] = tid[0], qid[0], ts[0], te[0], qs[0], qe[0]
        if seed % 3 == 0:
            o = np.argsort(qid, kind="stable")
            if o[0] == 0:                                           # keep record 0 first so the mirror still refers to it
                qid, qs, qe, tid, ts, te = (a[o] for a in (qid, qs, qe, tid, ts, te))
    L = int(rng.choice([60, 100, 250, 1000]))
    p = RaftParams(reso=reso, est_cov=int(rng.integers(1, 6)), cov_mul=float(rng.choice([1.0, 1.3, 1.5, 2.0])),
                   repeat_length=L, interval_length=L, read_length=L * int(rng.integers(1, 4)) + int(rng.integers(0, L)),
                   overlap_length=int(rng.integers(0, min(L, 60))), flanking_length=int(rng.integers(0, 300)))
    return p, [rl, qid, qs, qe, tid, ts, te]


@pytest.mark.parametrize("seed", range(40))
def test_random_small_vs_oracle(eng_mod, seed):
    p, cols = random_case(seed)
    want = oracle_run(p, *cols)
    for tile, bucket, variant in ((0, False, "wave"), (32, False, ("wave", "deep")[seed % 2]), (0, True, ("wave", "deep")[(seed + 1) % 2]), (200, False, "deep")):
        got, s = run_engine(eng_mod, p, cols, tile_bins=tile, force_bucket=bucket, variant=variant)
        assert_same_result(got, want, f"seed {seed} tile {tile} bucket {bucket} variant {variant}")


@pytest.mark.parametrize("variant", ["wave", "deep"])
@pytest.mark.parametrize("name", ["s60_ultralong", "s200_smallparams", "s300_default", "edge_reads"])
def test_golden_cases_all_kernel_variants(eng_mod, name, variant):
    p, cols, exp, meta = load_case(name)
    want = oracle_run(p, *cols)
    got, s = run_engine(eng_mod, p, cols, variant=variant)
    assert_same_result(got, want, f"{name}/variant {variant}")


@pytest.mark.parametrize("kw", [dict(n_reads=5000, seed=21), dict(n_reads=4000, seed=22, symmetric=False, shuffle=True),
                                dict(n_reads=400, seed=23, mean_len=120000, sigma=0.9, max_len=1500000, coverage=25,
                                     n_families=8, copies=5, rep_len=(20000, 60000))])
def test_synthetic_medium_vs_oracle(eng_mod, kw):
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)), reso=50 if kw["n_reads"] > 1000 else 10)
    want = oracle_run(p, *cols)
    got, s = run_engine(eng_mod, p, cols, device_resident=True)
    assert_same_result(got, want, str(kw))
    got, s = run_engine(eng_mod, p, cols, force_bucket=True)
    assert_same_result(got, want, str(kw) + " bucket")


def test_empty_inputs(eng_mod):
    p = RaftParams(est_cov=3)
    z = np.empty(0, np.int32)
    got, s = run_engine(eng_mod, p, [np.array([100, 0, 7], np.int32), z, z, z, z, z, z])
    want = oracle_run(p, np.array([100, 0, 7], np.int32), z, z, z, z, z, z)
    assert_same_result(got, want, "no records")
    got, s = run_engine(eng_mod, p, [z, z, z, z, z, z, z])
    assert s.n_bins == 0 and s.n_fragments == 0


def test_defined_errors_match_oracle(eng_mod):
    p = RaftParams(est_cov=2, reso=50, repeat_length=100, interval_length=100, read_length=200, overlap_length=20)
    rl = np.array([230, 100], np.int32)
    one = lambda *v: [np.array([x], np.int32) for x in v]
    cases = [(p, one(0, 0, 10, 2, 0, 10), eng_mod.ERR_READ_ID), (p, one(5, 0, 10, 1, 0, 10), eng_mod.ERR_READ_ID),
             (p, one(0, 0, 251, 1, 0, 10), eng_mod.ERR_COORD), (p, one(0, -4, 10, 1, 0, 10), eng_mod.ERR_COORD),
             (RaftParams(est_cov=2, reso=50, repeat_length=100, interval_length=100, read_length=200, overlap_length=250),
              one(0, 0, 10, 1, 0, 10), eng_mod.ERR_FRAGMENT)]
    for pp, cols, code in cases:
        with pytest.raises(OracleError) as oe:
            oracle_run(pp, rl, *cols)
        assert oe.value.code == code
        with pytest.raises(eng_mod.RaftError) as ge:
            run_engine(eng_mod, pp, [rl] + cols)
        assert ge.value.code == code
    got, _ = run_engine(eng_mod, p, [rl] + one(0, 0, 250, 1, 0, 10))   # beyond len but inside the last window: defined
    assert_same_result(got, oracle_run(p, rl, *one(0, 0, 250, 1, 0, 10)), "e>len inside last window")


def test_full_size_properties(eng_mod):
    """Config-2 scale (50 k reads, ~4.3 M records) through size-independent invariants, plus the oracle."""
    import torch

    from raft_amd.synth import make_overlaps
    o = make_overlaps(50000, seed=2, device="cuda:0")
    p = RaftParams(est_cov=30)
    eng = eng_mod.Engine(p, device=0)
    eng.use_torch_stream()
    eng.run_device(o.read_len, *o.columns())
    s = eng.finish()
    out = eng.outputs_device()
    nb = (o.read_len.long() + p.reso - 1) // p.reso
    assert s.n_bins == int(nb.sum()) and s.symmetric == 1 and s.interval_path == 0 and s.n_segments == 2
    # sum of coverage == sum over intervals of the number of windows they touch
    first = o.qs.long() // p.reso
    last = (o.qe.long() - 1) // p.reso
    assert s.total_coverage == int((last - first + 1).clamp(min=0).sum())
    assert int(out["cov"].long().sum()) == s.total_coverage
    assert s.total_read_length == int(o.read_len.long().sum())
    # fragments tile every read: first begins at 0, last ends at len, consecutive ones overlap by overlap_length
    fo, fb, fe, fr = out["frag_offset"], out["frag_begin"], out["frag_end"], out["frag_read"].long()
    assert bool((fb[fo[:-1]] == 0).all()) and bool((fe[fo[1:] - 1] == o.read_len).all())
    same = fr[1:] == fr[:-1]
    assert bool(((fe[:-1] - fb[1:])[same] == p.overlap_length).all())
    # same result from the bucketing path, and bit-exact vs the oracle
    a = eng.fetch()
    eng.set_tuning(0, True)
    eng.run_device(o.read_len, *o.columns())
    s2 = eng.finish()
    b = eng.fetch()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert s2.interval_path == 1 and s2.n_intervals == o.n_rec
    want = oracle_run(p, *[c.cpu().numpy() for c in (o.read_len,) + o.columns()])
    a.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
             total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    assert_same_result(a, want, "S50k")
    eng.close()


# ---- shapes that stress the tile paths (chosen for the workgroup-tile kernels of rounds 1-3; now through the wave and the deep kernel) ----

def _sym_case(rl, qid, s, e, rng):
    """Symmetric record set from query-side intervals: record 1 mirrors record 0, targets are arbitrary other reads."""
    n = len(rl)
    tid = ((qid.astype(np.int64) + 1 + rng.integers(0, max(n - 1, 1), len(qid))) % n).astype(np.int32)
    ts, te = np.zeros_like(s), np.minimum(rl[tid], 1).astype(np.int32)
    cols = [np.asarray(a, np.int32) for a in (qid, s, e, tid, ts, te)]
    # mirror of record 0 placed at position 1 (chop.hpp:175-184), kept consistent with the sorted order by choosing
    # the mirror's query (= record 0's target) freely: it only has to exist
    q0, s0, e0, t0, ts0, te0 = (int(c[0]) for c in cols)
    mir = [t0, ts0, te0, q0, s0, e0]
    return [np.insert(c, 1, v).astype(np.int32) for c, v in zip(cols, mir)]


def _intervals(rl, ids, rng, frac=1.0):
    ln = rl[ids].astype(np.int64)
    a = (rng.random(len(ids)) * ln).astype(np.int64)
    b = np.minimum(ln, a + 1 + (rng.random(len(ids)) * ln * frac).astype(np.int64))
    return a.astype(np.int32), b.astype(np.int32)


def _fast_kernel_cases():
    rng = np.random.default_rng(77)
    cases = {}
    # (a) very many tiny reads per tile (more than a tile's per-read table holds: 63) between ordinary reads
    rl = np.concatenate([rng.integers(2000, 30000, 40), rng.integers(60, 400, 1500), rng.integers(2000, 30000, 40)]).astype(np.int32)
    qid = np.sort(rng.integers(0, len(rl), 30000)).astype(np.int32)
    s, e = _intervals(rl, qid, rng)
    cases["tiny_reads"] = (RaftParams(est_cov=8), [rl] + _sym_case(rl, qid, s, e, rng))
    # (b) dense tiles: far more intervals per tile and sorted run than the prefetch slots hold
    rl = rng.integers(20000, 40000, 60).astype(np.int32)
    parts = [np.sort(rng.integers(0, len(rl), 40000)) for _ in range(2)]
    qid = np.concatenate(parts).astype(np.int32)
    s, e = _intervals(rl, qid, rng, 0.3)
    cases["dense_two_runs"] = (RaftParams(est_cov=100), [rl] + _sym_case(rl, qid, s, e, rng))
    # (c) one sorted run / four sorted runs (other instantiations of the kernel)
    rl = rng.integers(5000, 60000, 300).astype(np.int32)
    for name, k in (("one_run", 1), ("several_runs", 3)):   # the inserted mirror may add a run: still at most four
        qid = np.concatenate([np.sort(rng.integers(0, len(rl), 6000)) for _ in range(k)]).astype(np.int32)
        s, e = _intervals(rl, qid, rng, 0.6)
        cols = _sym_case(rl, qid, s, e, rng)
        if k == 1:                                   # keep the stream one ascending run: the mirror goes where it sorts
            o = np.argsort(cols[0][1:], kind="stable") + 1
            cols = [np.concatenate([c[:1], c[o]]) for c in cols]
            if cols[0][1] < cols[0][0]: cols = None
        if cols is not None: cases[name] = (RaftParams(est_cov=20), [rl] + cols)
    # (d) reads longer than the LDS window in the middle of ordinary ones (default reso: > 307 kb)
    rl = rng.integers(8000, 30000, 120).astype(np.int32)
    rl[[17, 60, 61]] = [450000, 330000, 900000]
    qid = np.concatenate([np.sort(rng.integers(0, len(rl), 9000)) for _ in range(2)]).astype(np.int32)
    s, e = _intervals(rl, qid, rng, 0.5)
    cases["long_reads_between"] = (RaftParams(est_cov=25), [rl] + _sym_case(rl, qid, s, e, rng))
    # (e) long runs across rows, waves and tiles with read starts inside them: every window high
    rl = rng.integers(9000, 45000, 200).astype(np.int32)
    qid = np.sort(np.repeat(np.arange(len(rl)), 12)).astype(np.int32)
    s = np.zeros(len(qid), np.int32); e = rl[qid].astype(np.int32)
    drop = rng.random(len(qid)) < 0.15                        # thin the coverage in places: runs end and begin
    s[drop] = (rl[qid[drop]] * 0.4).astype(np.int32); e[drop] = (rl[qid[drop]] * 0.6).astype(np.int32)
    cases["all_high"] = (RaftParams(est_cov=6), [rl] + _sym_case(rl, qid, s, e, rng))
    return cases


FAST_CASES = _fast_kernel_cases()


@pytest.mark.parametrize("variant", ["wave", "deep"])
@pytest.mark.parametrize("name", sorted(FAST_CASES))
def test_fast_kernel_paths(eng_mod, name, variant):
    p, cols = FAST_CASES[name]
    want = oracle_run(p, *cols)
    got, s = run_engine(eng_mod, p, cols, variant=variant)
    assert_same_result(got, want, f"{name}/variant {variant}")
    assert s.symmetric == 1 and s.interval_path == 0          # the sorted-segment path, i.e. the kernels under test
    if name == "all_high": assert want["rep_s"].size > 50     # the case does produce long repeats
