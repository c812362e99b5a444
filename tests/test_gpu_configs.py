"""GPU: BASELINE.json's configs at their own workloads, through the C ABI, bit-exact.

  config 1  (configs[0]) 2 Mbp / 42x stand-in, expected arrays parsed from the reference binary's files
  config 3  (configs[2]) the bench workload itself -- 3.3 M reads, 2.9e8 records -- by invariants, by the
            counting-sort path against the sorted-segment path, and by the oracle on windows of reads cut out of it
  config 5  (configs[4]) ultralong 60x / 150 kb reads with 50 kb tandem arrays: -p x -m x -r sweep vs the oracle
  plus      tests/golden/ref_fuzz.npz (320 random inputs, outputs of the reference BINARY) and the std::sort tie corner
"""
import numpy as np
import pytest
from raft_testlib import (RaftParams, assert_matches_ref_fuzz, assert_same_result, load_config1, oracle_run, ref_fuzz_case,
                          ref_fuzz_count, tie_case)

pytestmark = pytest.mark.gpu


def engine_result(eng, s, fetch_kw=None):
    got = eng.fetch(**(fetch_kw or {}))
    got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
               total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    return got


def run_host(p, cols, **tuning):
    """tuning: tile_bins, force_bucket, kernel ("wave" as it comes / "deep": every tile through pileup_deep_kernel)."""
    from raft_amd import engine
    from raft_testlib import kernel_mode
    eng = engine.Engine(p, device=0)
    try:
        if tuning:
            eng.set_tuning(tuning.get("tile_bins", 0), tuning.get("force_bucket", False), -1)
        with kernel_mode(tuning.get("kernel", "wave")):
            eng.run_host(*cols)
            s = eng.finish()
        return engine_result(eng, s), s
    finally:
        eng.close()


# ---- config 1 -----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("mode", ["auto", "bucket", "deep"])
def test_config1_standin_vs_reference_outputs(mode):
    p, cols, exp, meta = load_config1()
    got, s = run_host(p, cols, force_bucket=(mode == "bucket"), kernel="deep" if mode == "deep" else "wave")
    assert s.symmetric == meta["symmetric"] and s.n_records == meta["n_rec"]
    for k in exp:
        assert np.array_equal(got[k], exp[k]), (mode, k)
    assert "coverage per window is %f \n" % (s.total_coverage / s.total_windows) in meta["stdout"]
    assert "fraction_of_repeat_length %f \n" % (s.total_repeat_length / s.total_read_length) in meta["stdout"]


# ---- the reference binary's outputs on random inputs; the std::sort tie corner -----------------------------------------

def test_engine_vs_reference_binary_fuzz():
    assert ref_fuzz_count() >= 200
    from raft_amd import engine
    for i in range(ref_fuzz_count()):
        p, cols, exp = ref_fuzz_case(i)
        from raft_testlib import kernel_mode
        eng = engine.Engine(p, device=0)
        try:
            eng.set_tuning(0, i % 3 == 2, -1)            # (every third case: the general bucketing path; another third: the deep kernel)
            with kernel_mode("deep" if i % 3 == 1 else "wave"):
                eng.run_host(*cols)
                s = eng.finish()
            assert_matches_ref_fuzz(engine_result(eng, s), exp, p, f"ref_fuzz case {i}")
        finally:
            eng.close()


@pytest.mark.parametrize("seed", range(0, 150, 3))
def test_repeat_sort_ties_vs_oracle(seed):
    """repeat.hpp:170 with > 16 repeats of which several clamp to start 0: libstdc++'s order (finalize.hpp rep_std_sort)."""
    p, cols = tie_case(seed)
    want = oracle_run(p, *cols)
    got, s = run_host(p, cols)
    assert_same_result(got, want, f"tie case {seed}")


# ---- config 3 at full size ---------------------------------------------------------------------------------------------

def window_subproblem(o, a, b):
    """raft_amd.synth.query_window as numpy columns for the oracle."""
    from raft_amd.synth import query_window
    w = query_window(o, a, b)
    return w.read_len.cpu().numpy(), [c.cpu().numpy() for c in w.columns()]


def test_config3_full_size():
    """The bench workload (BASELINE configs[2]: 3.3 M reads, ~2.9e8 records): invariants, both interval paths, oracle on
    three 20 k-read windows cut from the full-size result."""
    import torch
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    o = make_overlaps(3_300_000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
    p = RaftParams(est_cov=32)
    eng = engine.Engine(p, device=0)
    eng.use_torch_stream()
    eng.run_device(o.read_len, *o.columns())
    s = eng.finish()
    out = eng.outputs_device()
    assert o.n_rec > 2.5e8 and s.n_records == o.n_rec and s.symmetric == 1 and s.interval_path == 0 and s.n_segments == 2
    nb = (o.read_len.long() + p.reso - 1) // p.reso
    assert s.n_bins == int(nb.sum()) == s.total_windows
    assert torch.equal(out["cov_offset"][1:], torch.cumsum(nb, 0))
    # sum of coverage == windows touched by the intervals == total_coverage
    touched = ((o.qe.long() - 1) // p.reso - o.qs.long() // p.reso + 1).clamp(min=0)
    assert s.total_coverage == int(touched.sum()) == int(out["cov"].sum(dtype=torch.int64))
    # per read too: coverage summed over a read's windows == windows touched by that read's intervals
    per_read = torch.zeros(o.n_reads, dtype=torch.int64, device="cuda:0").index_add_(0, o.qid.long(), touched)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda:0"), torch.cumsum(out["cov"], 0, dtype=torch.int64)])
    assert torch.equal(csum[out["cov_offset"][1:]] - csum[out["cov_offset"][:-1]], per_read)
    del csum, per_read, touched
    assert s.total_read_length == int(o.read_len.long().sum())
    # fragments tile every read with overlap_length back-overlap; read_num is dense (row index + 1)
    fo, fb, fe, fr = out["frag_offset"], out["frag_begin"], out["frag_end"], out["frag_read"].long()
    assert int(fo[-1]) == s.n_fragments >= o.n_reads
    assert bool((fb[fo[:-1]] == 0).all()) and bool((fe[fo[1:] - 1] == o.read_len).all())
    same = fr[1:] == fr[:-1]
    assert bool(((fe[:-1] - fb[1:])[same] == p.overlap_length).all()) and bool((fr[1:] >= fr[:-1]).all())
    assert bool(((fr[1:] - fr[:-1])[~same] == 1).all()) and bool((fe > fb).all())
    # repeats: inside the read, at least repeat_length long (a run's last window may be partial: up to reso - 1 bp less)
    ro, rs, re_ = out["rep_offset"], out["rep_s"], out["rep_e"]
    rr = torch.repeat_interleave(torch.arange(o.n_reads, device="cuda:0"), (ro[1:] - ro[:-1]))
    assert s.n_repeats == int(ro[-1]) > 1e5
    assert bool((rs >= 0).all()) and bool((re_ <= o.read_len[rr]).all()) and bool((re_ - rs > p.repeat_length - p.reso).all())
    # the oracle on three windows of 20 k reads cut out of the full-size result
    host = {k: v.cpu().numpy() for k, v in out.items() if k != "cov"}
    for a in (0, 1_640_000, 3_280_000):
        b = a + 20_000
        rl, cols = window_subproblem(o, a, b)
        want = oracle_run(p, rl, *cols)
        assert want["symmetric"] == 1
        n = b - a
        c0, c1 = int(host["cov_offset"][a]), int(host["cov_offset"][b])
        assert np.array_equal(out["cov"][c0:c1].cpu().numpy(), want["cov"][: int(want["cov_offset"][n])]), a
        for key, arrs in (("rep", ("rep_s", "rep_e")), ("cut", ("cuts",)), ("frag", ("frag_begin", "frag_end"))):
            off = host[key + "_offset"]
            assert np.array_equal(off[a:b + 1] - off[a], want[key + "_offset"][: n + 1]), (a, key)
            for k in arrs:
                assert np.array_equal(host[k][off[a]:off[b]], want[k][: int(want[key + "_offset"][n])]), (a, k)
        assert np.array_equal(host["frag_read"][host["frag_offset"][a]:host["frag_offset"][b]] - a,
                              want["frag_read"][: int(want["frag_offset"][n])])
    # the counting-sort path on the same set: identical outputs
    keep = {k: v.clone() for k, v in out.items()}
    tot = (s.n_bins, s.n_repeats, s.n_cuts, s.n_fragments, s.total_coverage, s.total_repeat_length, s.total_read_length)
    eng.set_tuning(0, True)
    eng.run_device(o.read_len, *o.columns())
    s2 = eng.finish()
    out2 = eng.outputs_device()
    assert s2.interval_path == 1 and s2.n_intervals == o.n_rec
    assert tot == (s2.n_bins, s2.n_repeats, s2.n_cuts, s2.n_fragments, s2.total_coverage, s2.total_repeat_length, s2.total_read_length)
    for k in keep:
        assert torch.equal(keep[k], out2[k]), k
    eng.close()


# ---- config 5: ultralong reads, tandem arrays, parameter sweep ---------------------------------------------------------------

@pytest.fixture(scope="module")
def ultralong_set():
    from raft_amd.synth import make_overlaps
    o = make_overlaps(500, mean_len=150000.0, sigma=0.7, min_len=10000, max_len=1_500_000, coverage=60.0, seed=55,
                      n_families=2, copies=6, rep_len=(45000, 55000))
    return [c.numpy() for c in (o.read_len,) + o.columns()]


@pytest.mark.parametrize("reso", [10, 50])
@pytest.mark.parametrize("cov_mul", [1.3, 1.5, 2.0])
@pytest.mark.parametrize("plen", [5000, 10000, 50000])
def test_config5_parameter_sweep(ultralong_set, plen, cov_mul, reso):
    """-p {5000,10000,50000} x -m {1.3,1.5,2.0} x -r {10,50} on a 60x / 150 kb-mean set with
    50 kb tandem arrays at 6 copies; -r 10 puts reads of up to 150 k windows through the chunked long-read path."""
    cols = ultralong_set
    p = RaftParams(reso=reso, est_cov=60, cov_mul=cov_mul, repeat_length=plen, interval_length=plen, read_length=2 * plen)
    want = oracle_run(p, *cols)
    assert want["rep_s"].size > 0 or cov_mul == 2.0
    got, s = run_host(p, cols)
    assert_same_result(got, want, f"-p {plen} -m {cov_mul} -r {reso}")
    if (plen, cov_mul) in ((5000, 1.3), (50000, 1.5)):     # every tile through the deep kernel, and the counting-sort path as well
        got, s = run_host(p, cols, kernel="deep")
        assert_same_result(got, want, f"-p {plen} -m {cov_mul} -r {reso} deep kernel")
        got, s = run_host(p, cols, force_bucket=True)
        assert_same_result(got, want, f"-p {plen} -m {cov_mul} -r {reso} counting sort")


# ---- the transfer encoding of cov[] and the three-column upload -----------------------------------------------------------

def test_packed_fetch_and_query_only_upload():
    """raft_hip_fetch_packed + raft_host_unpack_coverage reproduce raft_hip_fetch; with symmetric_mode = 1 the target
    columns are not passed at all.  Coverage >= 255 (exceptions) is forced by stacking records on a few reads."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(3000, seed=77, coverage=40.0)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    hot = np.flatnonzero(cols[1] == cols[1][len(cols[1]) // 2])[:1]
    extra = [np.repeat(c[hot], 700) for c in cols[1:]]                    # 700 copies of one record: cov >= 255 there
    cols = [cols[0]] + [np.concatenate([c, x]) for c, x in zip(cols[1:], extra)]
    p = RaftParams(est_cov=40)
    want = oracle_run(p, *cols)
    assert want["symmetric"] == 1 and int(want["cov"].max()) >= 700
    eng = engine.Engine(RaftParams(est_cov=40, symmetric_mode=1), device=0)
    eng.run_host(cols[0], cols[1], cols[2], cols[3], None, None, None)      # query side only
    s = eng.finish()
    full = engine_result(eng, s)
    assert_same_result(full, want, "symmetric_mode=1, three columns")
    pk = eng.fetch_packed()
    assert pk["cov8"].dtype == np.uint8 and pk["exc_index"].size == int((want["cov"] >= 255).sum()) > 0
    assert bool((np.diff(pk["exc_index"]) > 0).all())
    assert np.array_equal(hostio.unpack_coverage(pk["cov8"], pk["exc_index"], pk["exc_value"]), want["cov"])
    for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_read", "frag_begin", "frag_end"):
        assert np.array_equal(pk[k], want[k]), k
    pk2 = eng.fetch_packed(pinned=True, out=None)                           # a second call serves the cached encoding
    assert np.array_equal(pk2["cov8"], pk["cov8"])
    pk16 = eng.fetch_packed(width=2)                                        # two bytes per window: nothing reaches 65535 here
    assert pk16["cov8"].dtype == np.uint16 and pk16["exc_index"].size == 0
    assert np.array_equal(hostio.unpack_coverage(pk16["cov8"], pk16["exc_index"], pk16["exc_value"]), want["cov"])
    # chunked, one byte per window: the exceptions of every chunk land in the caller's list at the right place, also when
    # several contexts share the job (later contexts' exceptions are moved down behind the earlier ones')
    extra_hot = [np.repeat(c[len(c) // 7: len(c) // 7 + 1], 400) for c in cols[1:]]       # a second pile, on another read
    cols2 = [cols[0]] + [np.concatenate([c, x]) for c, x in zip(cols[1:], extra_hot)]
    order = np.argsort(cols2[1], kind="stable")                                           # keep the stream one sorted run
    cols2 = [cols2[0]] + [c[order] for c in cols2[1:]]
    want2 = oracle_run(p, *cols2)
    others = [engine.Engine(RaftParams(est_cov=40, symmetric_mode=1), device=0) for _ in range(2)]
    for oth, nch in (([], 5), (others[:1], 6), (others, 9)):
        r8, s8 = eng.run_pipelined(cols2[0], cols2[1], cols2[2], cols2[3], n_chunks=nch, others=oth)
        n_want = int((want2["cov"] >= 255).sum())
        assert n_want > 100 and r8["exc_index"].size == n_want, (r8["exc_index"].size, n_want, len(oth), nch)
        assert bool((np.diff(r8["exc_index"]) > 0).all()), (len(oth), nch)
        assert np.array_equal(hostio.unpack_coverage(r8["cov8"], r8["exc_index"], r8["exc_value"]), want2["cov"]), (len(oth), nch)
        for k in ("rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
            assert np.array_equal(r8[k], want2[k]), (k, len(oth), nch)
    for e2 in others:
        e2.close()
    res16, s16 = eng.run_pipelined(cols[0], cols[1], cols[2], cols[3], n_chunks=4,
                                   out=eng.host_output_buffers(cols[0], pinned=False, width=2))
    assert res16["cov8"].dtype == np.uint16
    assert np.array_equal(hostio.unpack_coverage(res16["cov8"], res16["exc_index"], res16["exc_value"]), want["cov"])
    eng.close()
    with pytest.raises(engine.RaftError):                                   # detection mode needs the target columns
        e2 = engine.Engine(RaftParams(est_cov=40), device=0)
        try:
            e2.run_host(cols[0], cols[1], cols[2], cols[3], None, None, None)
        finally:
            e2.close()


# ---- chunked host pipeline (raft_hip_run_pipelined) -----------------------------------------------------------------------

def check_pipelined(res, s, want, what):
    from raft_amd import hostio
    assert np.array_equal(hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"]), want["cov"]), what
    for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
        assert np.array_equal(res[k], want[k]), (what, k)
    assert (s.symmetric, s.high_cov, s.total_coverage, s.total_windows, s.total_repeat_length, s.total_read_length) == \
        tuple(want[k] for k in ("symmetric", "high_cov", "total_coverage", "total_windows", "total_repeat_length", "total_read_length")), what
    assert s.n_fragments == len(want["frag_read"]) and s.n_repeats == len(want["rep_s"]) and s.n_cuts == len(want["cuts"]), what
    assert s.n_intervals == want["n_intervals"], what


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
@pytest.mark.parametrize("kw,n_chunks", [(dict(n_reads=4000, seed=81), 5), (dict(n_reads=4000, seed=81), 2), (dict(n_reads=4000, seed=81), 23),
                                         (dict(n_reads=1500, seed=82, mean_len=90000, sigma=0.9, max_len=1_200_000, coverage=25), 4),
                                         (dict(n_reads=50000, seed=2), 0)])
def test_pipelined_equals_oracle(kw, n_chunks, n_ctx):
    """Upload / pass / download of consecutive read ranges overlapped: same outputs as the one-piece pass and the oracle.
    n_chunks = 0 lets the engine choose (one piece for the 50 k-read set: the threshold is 2^24 records)."""
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    o = make_overlaps(**kw)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(kw.get("coverage", 30)))
    want = oracle_run(p, *cols)
    eng = engine.Engine(RaftParams(**dict(p.__dict__, symmetric_mode=1)), device=0)
    # more contexts share the job (raft_hip_run_multi): on this box they sit on the same GPU; their own parameters are
    # overwritten by the first context's
    others = [engine.Engine(RaftParams(est_cov=3, reso=7), device=0) for _ in range(n_ctx - 1)]
    out = eng.host_output_buffers(cols[0], pinned=True)
    for rep in range(2):                                  # a second pass reuses lanes and buffers
        res, s = eng.run_pipelined(cols[0], cols[1], cols[2], cols[3], n_chunks=n_chunks, out=out, others=others)
        check_pipelined(res, s, want, f"{kw} chunks {n_chunks} contexts {n_ctx} pass {rep}")
        assert s.n_segments == 2
    if n_chunks:                                          # the context itself holds no pass after a chunked run
        with pytest.raises(engine.RaftError) as e:
            eng.fetch()
        assert e.value.code == engine.ERR_STATE
    for e2 in [eng] + others:
        e2.close()


def test_pipelined_shapes_and_fallbacks():
    from raft_amd import engine
    rng = np.random.default_rng(91)
    rl = rng.integers(3000, 40000, 900).astype(np.int32)

    def sym_set(k_runs):
        qid = np.concatenate([np.sort(rng.integers(0, len(rl), 5000)) for _ in range(k_runs)]).astype(np.int32)
        a = (rng.random(len(qid)) * rl[qid]).astype(np.int32)
        b = np.minimum(rl[qid], a + 1 + (rng.random(len(qid)) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
        return qid, a, b
    p1 = RaftParams(est_cov=12, symmetric_mode=1)
    eng = engine.Engine(p1, device=0)
    for k_runs in (1, 3, 4, 6):                          # 6 sorted runs: more than the plan accepts -> one-piece fallback
        qid, a, b = sym_set(k_runs)
        want = oracle_run(RaftParams(est_cov=12), rl, qid, a, b, qid, a, b)      # self overlaps: query side only either way
        res, s = eng.run_pipelined(rl, qid, a, b, n_chunks=4)
        want["symmetric"] = 1
        check_pipelined(res, s, want, f"{k_runs} runs")
    qid, a, b = sym_set(2)
    perm = rng.permutation(len(qid))                      # unsorted stream: the samples see many descents -> fallback
    want = oracle_run(RaftParams(est_cov=12), rl, qid[perm], a[perm], b[perm], qid[perm], a[perm], b[perm]); want["symmetric"] = 1
    res, s = eng.run_pipelined(rl, qid[perm], a[perm], b[perm], n_chunks=4)
    check_pipelined(res, s, want, "shuffled")
    # a stream that looks sorted to the samples but is not: two far-apart records swapped.  The chunk that receives a
    # foreign read id reports it and the job is redone in one piece -- results still exact.
    q2, a2, b2 = qid.copy(), a.copy(), b.copy()
    i, j = 7, len(q2) // 2 - 11
    for arr in (q2, a2, b2):
        arr[i], arr[j] = arr[j], arr[i]
    want = oracle_run(RaftParams(est_cov=12), rl, q2, a2, b2, q2, a2, b2); want["symmetric"] = 1
    res, s = eng.run_pipelined(rl, q2, a2, b2, n_chunks=6)
    check_pipelined(res, s, want, "swapped records")
    # data errors come back as from run_host: same code, same index
    bad = b.copy(); bad[1234] = rl[qid[1234]] + 5000
    with pytest.raises(engine.RaftError) as e1:
        eng.run_pipelined(rl, qid, a, bad, n_chunks=4)
    with pytest.raises(engine.RaftError) as e2:
        eng.run_host(rl, qid, a, bad, None, None, None); eng.finish()
    assert e1.value.code == e2.value.code == engine.ERR_COORD and e1.value.index == e2.value.index == 1234
    # capacities: too small -> defined error
    small = eng.host_output_buffers(rl, pinned=False)
    small["cov8"] = small["cov8"][: small["cov8"].size // 2]
    with pytest.raises(engine.RaftError) as e3:
        eng.run_pipelined(rl, qid, a, b, n_chunks=4, out=small)
    assert e3.value.code == engine.ERR_TOO_LARGE
    # detection mode (symmetric_mode = -1) and empty inputs take the one-piece path
    eng.close()
    eng = engine.Engine(RaftParams(est_cov=12), device=0)
    want = oracle_run(RaftParams(est_cov=12), rl, qid, a, b, qid, a, b)
    res, s = eng.run_pipelined(rl, qid, a, b, qid, a, b, n_chunks=4)
    check_pipelined(res, s, want, "detection mode")
    z = np.empty(0, np.int32)
    res, s = eng.run_pipelined(rl, z, z, z, z, z, z)
    assert s.n_fragments >= len(rl) and int(res["cov8"].sum()) == 0
    eng.close()


def test_misspeculated_pass_is_redone():
    """A pass starts from a sampled guess of the sorted runs while inspect_kernel looks at every record (engine.hip
    run_pass).  Streams the samples misjudge -- disorder between two samples, a read id out of range between two samples,
    a detecting context meeting a non-symmetric PAF -- must come out exactly as without speculation."""
    from raft_amd import engine
    rng = np.random.default_rng(123)
    rl = rng.integers(3000, 40000, 2000).astype(np.int32)
    n = 60000                                             # > 16 k samples: most records are never sampled
    qid = np.sort(rng.integers(0, len(rl), n)).astype(np.int32)
    a = (rng.random(n) * rl[qid]).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.5).astype(np.int32)).astype(np.int32)
    p = RaftParams(est_cov=12)
    eng = engine.Engine(RaftParams(est_cov=12, symmetric_mode=1), device=0)
    for trial in range(6):
        q2, a2, b2 = qid.copy(), a.copy(), b.copy()
        i, j = sorted(rng.integers(1, n - 1, 2).tolist())
        for arr in (q2, a2, b2):
            arr[i], arr[j] = arr[j], arr[i]               # two far-apart records swapped: two hidden dips in the order
        want = oracle_run(p, rl, q2, a2, b2, q2, a2, b2); want["symmetric"] = 1
        eng.run_host(rl, q2, a2, b2, None, None, None)
        s = eng.finish()
        assert_same_result(engine_result(eng, s), want, f"swap {i} <-> {j}")
    q3 = qid.copy(); q3[n // 3 + 5] = len(rl) + 7         # an id out of range that no sample sees
    with pytest.raises(engine.RaftError) as e:
        eng.run_host(rl, q3, a, b, None, None, None); eng.finish()
    assert e.value.code == engine.ERR_READ_ID and e.value.index == n // 3 + 5
    eng.close()
    # detection mode: the first pass assumes a symmetric PAF, is refuted, and the context remembers the answer
    tid = ((qid.astype(np.int64) + 1) % len(rl)).astype(np.int32)
    ts = np.zeros(n, np.int32); te = np.minimum(rl[tid], 100).astype(np.int32)
    want = oracle_run(p, rl, qid, a, b, tid, ts, te)
    assert want["symmetric"] == 0
    eng = engine.Engine(p, device=0)
    for rep in range(2):
        eng.run_host(rl, qid, a, b, tid, ts, te)
        s = eng.finish()
        assert_same_result(engine_result(eng, s), want, f"non-symmetric, pass {rep}")
    eng.close()


@pytest.mark.parametrize("tile_bins", [0, 256, 1024])
def test_ids_beyond_the_last_read_at_a_run_tail(tile_bins):
    """A verified pass (no inspect_kernel) must not lose the records behind the last read's: ids >= n_reads sort to the
    tail of a run, where the only tile boundary left is the closing one -- a tile in which no read begins owns them, and
    no kernel walks such a tile.  They have to refute the pass (tile_desc_kernel), so that the redo reports
    RAFT_HIP_ERR_READ_ID with the first such record, in every form of the pass: symmetric flag handed over, detecting
    context, chunked pipeline."""
    from raft_amd import engine
    rng = np.random.default_rng(2024 + tile_bins)
    rl = rng.integers(3000, 40000, 1200).astype(np.int32)
    rl[-1] = 39999                                        # the last read crosses tile boundaries: read-less tiles behind it
    n = 50000
    qid = np.sort(rng.integers(0, len(rl), n)).astype(np.int32)
    tid = rng.integers(0, len(rl), n).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.8).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.2).astype(np.int32)).astype(np.int32)
    ta = (rng.random(n) * rl[tid] * 0.8).astype(np.int32)
    tb = np.minimum(rl[tid], ta + 1 + (rng.random(n) * rl[tid] * 0.2).astype(np.int32)).astype(np.int32)
    order = np.argsort(tid, kind="stable")               # mirrored records, sorted by their query: the second run
    sym = [np.concatenate([x, y[order]]) for x, y in ((qid, tid), (a, ta), (b, tb), (tid, qid), (ta, a), (tb, b))]
    bad = len(rl) + 7
    for where in ("first run", "second run"):
        cols = [c.copy() for c in sym]
        at = n - 3 if where == "first run" else 2 * n - 3   # the last three records of the run name a read that does not exist
        cols[0][at:at + 3] = bad
        for mode in (1, -1):
            eng = engine.Engine(RaftParams(est_cov=12, symmetric_mode=mode), device=0)
            eng.set_tuning(tile_bins, False)
            with pytest.raises(engine.RaftError) as e:
                if mode == 1:
                    eng.run_host(rl, cols[0], cols[1], cols[2], None, None, None)
                else:
                    eng.run_host(rl, *cols)
                eng.finish()
            assert e.value.code == engine.ERR_READ_ID and e.value.index == at, (where, mode, e.value.code, e.value.index)
            if mode == 1:
                for n_chunks in (2, 5):
                    with pytest.raises(engine.RaftError) as e:
                        eng.run_pipelined(rl, cols[0], cols[1], cols[2], n_chunks=n_chunks)
                    assert e.value.code == engine.ERR_READ_ID and e.value.index == at, (where, n_chunks, e.value.index)
            eng.close()


def test_chunked_pipeline_with_records_out_of_place():
    """symmetric_mode = 1 on a stream that is NOT sorted: single records of late reads planted in early parts of a run.  The
    host's chunk cuts (bisections of the qid column) may or may not land on them; whatever they do, a record that ends up
    in a chunk whose reads it does not belong to has to send the job to the one-piece pass: results equal the oracle."""
    from raft_amd import engine
    rng = np.random.default_rng(99)
    rl = rng.integers(3000, 40000, 1500).astype(np.int32)
    n = 40000
    qid = np.sort(rng.integers(0, len(rl), n)).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.8).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.2).astype(np.int32)).astype(np.int32)
    p = RaftParams(est_cov=12)
    eng = engine.Engine(RaftParams(est_cov=12, symmetric_mode=1), device=0)
    for trial in range(8):
        q2, a2, b2 = qid.copy(), a.copy(), b.copy()
        for _ in range(1 + trial % 3):
            src = int(rng.integers(n * 3 // 4, n))           # a record of a late read ...
            dst = int(rng.integers(0, n // 2))               # ... moved into the early part of the run
            for arr in (q2, a2, b2):
                v = arr[src]
                arr[dst + 1:src + 1] = arr[dst:src].copy()
                arr[dst] = v
        want = oracle_run(p, rl, q2, a2, b2, q2, a2, b2); want["symmetric"] = 1
        res, s = eng.run_pipelined(rl, q2, a2, b2, n_chunks=2 + trial)
        check_pipelined(res, s, want, f"trial {trial}")
    eng.close()


def test_detecting_context_alternating_inputs():
    """A detecting context assumes a symmetric PAF and has tile_desc_kernel look for the mirror of record 0 (pileup.hpp
    MirrorArgs); after a PAF without one it stops assuming until a pass of its own has detected one again.  Symmetric and
    non-symmetric inputs in turn on ONE context: every pass equals the oracle, whichever way it was run."""
    from raft_amd import engine
    rng = np.random.default_rng(77)
    rl = rng.integers(3000, 40000, 1500).astype(np.int32)
    n = 40000
    qid = np.sort(rng.integers(0, len(rl), n)).astype(np.int32)
    tid = rng.integers(0, len(rl), n).astype(np.int32)
    a = (rng.random(n) * rl[qid] * 0.8).astype(np.int32)
    b = np.minimum(rl[qid], a + 1 + (rng.random(n) * rl[qid] * 0.2).astype(np.int32)).astype(np.int32)
    ta = (rng.random(n) * rl[tid] * 0.8).astype(np.int32)
    tb = np.minimum(rl[tid], ta + 1 + (rng.random(n) * rl[tid] * 0.2).astype(np.int32)).astype(np.int32)
    order = np.argsort(tid, kind="stable")               # the mirrored records, sorted by THEIR query: a second sorted run
    sym = tuple(np.concatenate([x, y[order]]) for x, y in ((qid, tid), (a, ta), (b, tb), (tid, qid), (ta, a), (tb, b)))
    nonsym = (qid, a, b, tid, ta, tb)
    p = RaftParams(est_cov=10)
    want = {"sym": oracle_run(p, rl, *sym), "nonsym": oracle_run(p, rl, *nonsym)}
    assert want["sym"]["symmetric"] == 1 and want["nonsym"]["symmetric"] == 0
    eng = engine.Engine(p, device=0)
    try:
        for step, which in enumerate(["sym", "sym", "nonsym", "nonsym", "sym", "sym", "nonsym", "sym"]):
            eng.run_host(rl, *(sym if which == "sym" else nonsym))
            s = eng.finish()
            assert_same_result(engine_result(eng, s), want[which], f"step {step}: {which}")
    finally:
        eng.close()


# ---- reads longer than a tile (pieces), runs of tiny reads (groups of up to 63) --------------------------------------------------

def _long_read_set(seed):
    """Short reads with a few reads far longer than the LDS window between them (default reso: > 397 kb), coverage thin in
    places and thick in others so that runs begin, end and cross piece boundaries; hundreds of tiny reads in a row."""
    rng = np.random.default_rng(seed)
    rl = rng.integers(5000, 40000, 260).astype(np.int32)
    rl[[3, 90, 91, 200]] = [1_450_000, 820_000, 2_100_000, 400_050]
    rl = np.concatenate([rl, rng.integers(0, 300, 400).astype(np.int32), rng.integers(5000, 40000, 30).astype(np.int32)])
    parts = []
    for _ in range(2):
        q = np.sort(np.concatenate([rng.integers(0, len(rl), 20000), np.repeat([3, 90, 91, 200], 4000)])).astype(np.int32)
        parts.append(q)
    qid = np.concatenate(parts)
    ln = rl[qid].astype(np.int64)
    a = (rng.random(len(qid)) * ln).astype(np.int64)
    b = np.minimum(ln, a + 1 + (rng.random(len(qid)) * np.minimum(ln, 60000)).astype(np.int64))
    return rl, qid, a.astype(np.int32), b.astype(np.int32)


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("mode", ["default", "deep"])
def test_recut_tiles_and_pieces(seed, mode, monkeypatch):
    """Reads longer than a tile go in pieces, runs of tiny reads in groups of up to 63 (the workers of pileup_wave_kernel cut them
    themselves; rounds 1-3: tile_desc_kernel re-cut them for the workgroup-tile kernel).  Through the wave kernel and through the deep
    kernel: both must equal the oracle, incl. runs crossing piece boundaries and the repeat-length total."""
    from raft_amd import engine
    if mode == "deep":
        monkeypatch.setenv("RAFT_DEEP_MIN", "1")
    rl, qid, a, b = _long_read_set(seed)
    for p in (RaftParams(est_cov=14, symmetric_mode=1), RaftParams(est_cov=9, reso=37, repeat_length=3000, interval_length=3000,
                                                                  read_length=9000, flanking_length=5000, symmetric_mode=1)):
        want = oracle_run(RaftParams(**dict(p.__dict__, symmetric_mode=-1)), rl, qid, a, b, qid, a, b)
        want["symmetric"] = 1
        assert want["rep_s"].size > 20
        eng = engine.Engine(p, device=0)
        try:
            eng.run_host(rl, qid, a, b, None, None, None)
            s = eng.finish()
            assert_same_result(engine_result(eng, s), want, f"seed {seed} {mode} reso {p.reso}")
        finally:
            eng.close()


@pytest.mark.parametrize("mode", ["default", "deep"])
def test_groups_of_reads_without_windows_are_still_checked(mode, monkeypatch):
    """A tile of nothing but zero-length reads: its (valid, empty) records pass, a record that needs a window is the
    COORD error the oracle reports -- whichever kernel the tiles are left to."""
    from raft_amd import engine
    if mode == "deep":
        monkeypatch.setenv("RAFT_DEEP_MIN", "1")
    from raft_testlib import OracleError
    rl = np.concatenate([np.full(300, 0, np.int32), np.array([5000, 0, 0, 7000], np.int32)])
    qid = np.sort(np.concatenate([np.arange(300), [300, 303, 303]])).astype(np.int32)
    s_ = np.zeros(len(qid), np.int32); e_ = np.zeros(len(qid), np.int32)
    e_[qid == 300] = 5000; e_[qid == 303] = 6000
    p = RaftParams(est_cov=1, symmetric_mode=1)
    want = oracle_run(RaftParams(est_cov=1), rl, qid, s_, e_, qid, s_, e_); want["symmetric"] = 1
    eng = engine.Engine(p, device=0)
    eng.run_host(rl, qid, s_, e_, None, None, None)
    assert_same_result(engine_result(eng, eng.finish()), want, "zero-length reads")
    bad = e_.copy(); bad[7] = 3                          # an interval on a read that has no window
    with pytest.raises(OracleError):
        oracle_run(RaftParams(est_cov=1), rl, qid, s_, bad, qid, s_, bad)
    with pytest.raises(engine.RaftError) as err:
        eng.run_host(rl, qid, s_, bad, None, None, None); eng.finish()
    assert err.value.code == engine.ERR_COORD
    eng.close()


def test_config4_two_shards_of_the_full_size_set():
    """BASELINE configs[3] at its own size on the one GPU of the box: the ONE 3.3 M-read set of configs[2] cut into two
    contiguous read ranges (raft_amd.dist.partition_reads), every shard handed the records of its reads in grouped form
    (what a rank of `bench.py --strong` runs).  The shards' totals add up to the single pass's; coverage, repeats and
    fragments of a 20 k-read window inside each shard equal the oracle's."""
    import torch
    from raft_amd import dist as rdist
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(3_300_000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
    p = RaftParams(est_cov=32)
    e0 = engine.Engine(p, device=0)
    e0.run_device(o.read_len, *o.columns())
    s0 = e0.finish()
    e0.close()
    ipr = torch.bincount(o.qid.long(), minlength=o.n_reads)
    bounds = rdist.partition_reads(o.read_len, p.reso, 2, intervals_per_read=ipr)
    tot = {k: 0 for k in ("n_bins", "n_repeats", "n_fragments", "total_coverage", "total_repeat_length", "total_read_length", "n_intervals")}
    eng = engine.Engine(RaftParams(est_cov=32, symmetric_mode=1), device=0)
    for g in range(2):
        b0, b1 = int(bounds[g]), int(bounds[g + 1])
        sel = (o.qid >= b0) & (o.qid < b1)
        q, qs, qe = (o.qid[sel] - b0).contiguous(), o.qs[sel].contiguous(), o.qe[sel].contiguous()
        rl = o.read_len[b0:b1].contiguous()
        off = hostio.group_offsets(b1 - b0, q.cpu().numpy())
        assert off is not None and off.shape[0] == 2
        B = int(((rl.long() + p.reso - 1) // p.reso).sum())
        eng.run_device_grouped(rl, torch.as_tensor(off).to("cuda:0"), q, qs, qe, n_bins=B)
        s = eng.finish()
        assert 0.4 < s.n_records / o.n_rec < 0.6 and s.interval_path == 0
        for k in tot:
            tot[k] += getattr(s, k)
        out = eng.outputs_device()
        host = {k: v.cpu().numpy() for k, v in out.items() if k != "cov"}
        a = b0 + (b1 - b0) // 2                           # a window in the middle of the shard (global read indices)
        rlw, cols = window_subproblem(o, a, a + 20_000)
        want = oracle_run(p, rlw, *cols)
        la, n = a - b0, 20_000
        c0, c1 = int(host["cov_offset"][la]), int(host["cov_offset"][la + n])
        assert np.array_equal(out["cov"][c0:c1].cpu().numpy(), want["cov"][: int(want["cov_offset"][n])]), g
        for key, arrs in (("rep", ("rep_s", "rep_e")), ("frag", ("frag_begin", "frag_end"))):
            offk = host[key + "_offset"]
            assert np.array_equal(offk[la:la + n + 1] - offk[la], want[key + "_offset"][: n + 1]), (g, key)
            for k in arrs:
                assert np.array_equal(host[k][offk[la]:offk[la + n]], want[k][: int(want[key + "_offset"][n])]), (g, k)
        del out, host
    eng.close()
    for k in tot:
        assert tot[k] == getattr(s0, k), k
