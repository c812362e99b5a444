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


def run_engine(engine, p, cols, tile_bins=0, force_bucket=False, device_resident=False, variant=-1):
    eng = engine.Engine(p, device=0)
    try:
        eng.set_tuning(tile_bins, force_bucket, variant)
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
    for tile, bucket, variant in ((0, False, -1), (32, False, seed % 4), (0, True, (seed + 2) % 4), (200, False, (seed + 3) % 4)):
        got, s = run_engine(eng_mod, p, cols, tile_bins=tile, force_bucket=bucket, variant=variant)
        assert_same_result(got, want, f"seed {seed} tile {tile} bucket {bucket} variant {variant}")


@pytest.mark.parametrize("variant", range(4))
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
