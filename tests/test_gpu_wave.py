"""GPU: the wave kernel (pileup_wave.hpp: one wave per tile, 16-bit difference array, workers that cut their own tiles) -- what is
specific to it, beyond the parity, configuration and consistency suites that run with it as the default pileup:

* against the 32-bit side kernel (pileup_deep.hpp, every tile sent its way) on shapes that stress its tile cutting: very many short
  reads (63 per tile), reads longer than a tile (pieces), dense stretches (tiles shrink to the reads whose records are in the
  slots; records behind the slots are streamed), a stream of four runs (columns only), reads without records or windows;
* the 16-bit bound: a tile with 2^15 or more intervals is left to pileup_deep_kernel in the same pass (tests/test_gpu_deep.py) --
  same arrays as the oracle;
* cut points written by the pass itself (raft_hip_set_emit_cuts) or by the first fetch: the same;
* the host pipeline deriving offsets and window records from the plain columns chunk by chunk (raft_hip_run_multi,
  symmetric_mode = 1): the oracle's arrays, and every input the derivation gives up on (ids out of place, a negative
  coordinate, a read beyond 65,535 windows) still ends as the one-piece pass ends.
Reference semantics: repeat.hpp:28-79, :111-168; chop.hpp:225-246.
"""
import numpy as np
import pytest
from raft_testlib import RaftParams, assert_same_result, oracle_run

pytestmark = pytest.mark.gpu


def engine_result(eng, cols):
    eng.run_host(*cols)
    s = eng.finish()
    got = eng.fetch()
    got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
               total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    return got, s


SHAPES = [
    dict(n_reads=30_000, mean_len=1200.0, coverage=15.0, seed=201, min_len=100, sigma=0.8),        # ~150 reads per 4096 windows: the 63-read limit
    dict(n_reads=3_000, mean_len=400_000.0, coverage=30.0, seed=202, min_len=50_000, max_len=2_000_000, sigma=0.6),   # pieces
    dict(n_reads=8_000, mean_len=25_000.0, coverage=300.0, seed=203),                              # dense: records far beyond the slots
    dict(n_reads=20_000, mean_len=20_000.0, coverage=30.0, seed=204, n_families=2000, copies=5),
]


@pytest.mark.parametrize("si", range(len(SHAPES)))
@pytest.mark.parametrize("reso", [50, 7])
def test_wave_kernel_equals_deep_kernel(si, reso):
    import torch
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    if reso == 7 and si == 1:
        pytest.skip("2 Mb reads at reso 7 exceed a window record's 16 bits: covered with columns in test_gpu_configs")
    o = make_overlaps(device="cuda:0", **SHAPES[si])
    p = RaftParams(est_cov=int(SHAPES[si]["coverage"]), reso=reso)
    cols = (o.read_len,) + o.columns()
    ref = None
    from raft_testlib import kernel_mode
    for variant, bucket in (("deep", False), ("wave", False), ("wave", True)):
        eng = engine.Engine(p, device=0)
        try:
            eng.set_tuning(0, bucket, -1)
            with kernel_mode(variant):
                eng.run_device(*cols)
                s = eng.finish()
            out = {k: v.clone() for k, v in eng.outputs_device().items()}
            tot = (s.symmetric, s.n_bins, s.n_repeats, s.n_cuts, s.n_fragments, s.total_coverage, s.total_repeat_length, s.total_read_length)
        finally:
            eng.close()
        if ref is None:
            ref = (out, tot)
            continue
        assert tot == ref[1], (SHAPES[si], variant, bucket)
        for k in out:
            assert torch.equal(out[k], ref[0][k]), (SHAPES[si], variant, bucket, k)


def test_wave_kernel_four_runs_and_window_records():
    """Columns in four sorted runs (a PAF concatenated from two pairs of files), and the same stream as two runs of window records."""
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(6000, seed=31)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    n = len(cols[1])
    cut = [0, n // 5, o.n_cis, (o.n_cis + n) // 2, n]          # each part stays sorted by query id: four runs
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    eng = engine.Engine(p, device=0)
    eng.set_tuning(0, False, 5)
    got, s = engine_result(eng, cols)
    assert_same_result(got, want, "two runs, columns")
    # four runs: split each file in two and interleave the halves' order inside the files
    order = np.concatenate([np.arange(cut[1], cut[2]), np.arange(cut[0], cut[1]), np.arange(cut[3], cut[4]), np.arange(cut[2], cut[3])])
    c4 = [cols[0]] + [c[order] for c in cols[1:]]
    want4 = oracle_run(p, *c4)
    got4, s4 = engine_result(eng, c4)
    assert s4.n_segments in (3, 4) and s4.interval_path == 0      # (more runs than two: the four-run instantiation)
    assert_same_result(got4, want4, "four runs, columns")
    eng.close()
    ps = RaftParams(est_cov=30, symmetric_mode=1)
    eng = engine.Engine(ps, device=0)
    eng.set_tuning(0, False, 5)
    off = hostio.group_offsets(o.n_reads, cols[1])
    win = hostio.pack_windows(cols[2], cols[3], 50)
    for width in (4, 1, 8):
        eng.set_output_width(width)
        eng.run_host_windows(cols[0], off, win)
        s = eng.finish()
        g = eng.fetch()
        g.update(symmetric=1, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                 total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
        assert_same_result(g, want, f"window records, width {width}")
    eng.close()


@pytest.mark.parametrize("two_runs", [False, True])
def test_records_behind_the_slots_stream_in_groups(two_runs):
    """A tile holds 256 records of a run in its prefetch slots (320 as window records); what lies behind them streams in groups of
    four batches of 64 (pileup_wave.hpp `cur.more`).  One read per tile, with record counts on every side of the slots' end, of a
    batch's and of a group's: columns and window records, one run and two, against the oracle."""
    from raft_amd import engine, hostio
    rng = np.random.default_rng(606 + two_runs)
    edges = [0, 1, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 258, 300, 319, 320, 321, 383, 384, 385, 511, 512, 513, 575, 576, 577,
             639, 640, 641, 767, 768, 769, 831, 832, 833, 1023, 1024, 1025, 1100, 1279, 1280, 1281, 1500]
    counts = edges + [int(x) for x in rng.integers(0, 1400, 40)]
    n = len(counts)
    read_len = rng.integers(150_000, 200_000, n).astype(np.int32)        # 3000 .. 4000 windows: one read per tile
    runs = []
    for r in range(2 if two_runs else 1):
        k = counts if r == 0 else list(reversed(counts))
        qid = np.repeat(np.arange(n, dtype=np.int32), k)
        L = read_len[qid].astype(np.int64)
        a, b = (rng.random(qid.size) * L).astype(np.int64), (rng.random(qid.size) * L).astype(np.int64)
        runs.append((qid, np.minimum(a, b).astype(np.int32), np.minimum(np.maximum(a, b) + 1, L).astype(np.int32)))
    qid, qs, qe = (np.concatenate([r[k] for r in runs]) for k in range(3))
    cols = [read_len, qid, qs, qe, qid.copy(), qs.copy(), qe.copy()]     # (target = query: nothing beside the query sides to pile up)
    p = RaftParams(est_cov=30, symmetric_mode=1)
    want = oracle_run(RaftParams(est_cov=30), *cols)
    eng = engine.Engine(p, device=0)
    try:
        got, s = engine_result(eng, cols[:4] + [None, None, None])
        got["symmetric"] = want["symmetric"]
        assert s.interval_path == 0 and s.n_segments == (2 if two_runs else 1)
        assert_same_result(got, want, "columns")
        off = hostio.group_offsets(n, qid)
        win = hostio.pack_windows(qs, qe, 50)
        for width in (4, 1, 2, 8):                   # (2: the two-byte encoding's 8-byte stores with a scalar offset, on every tile's edge rows)
            eng.set_output_width(width)
            eng.run_host_windows(read_len, off, win)
            s = eng.finish()
            g = eng.fetch()
            g.update(symmetric=want["symmetric"], high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                     total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
            assert_same_result(g, want, f"window records, width {width}")
    finally:
        eng.close()


def test_a_tile_too_deep_for_sixteen_bits_goes_to_the_deep_kernel():
    """40,000 intervals on one read: the wave kernel lists the tile and pileup_deep_kernel piles it up in the same pass -- coverage
    40,000 in the middle of the read, as the oracle has it."""
    from raft_amd import engine
    rng = np.random.default_rng(9)
    rl = np.array([30000, 12000, 50000, 8000], np.int32)
    m = 40000
    qid = np.concatenate([np.zeros(50, np.int32), np.full(m, 2, np.int32), np.full(30, 3, np.int32)])
    qs = np.zeros(qid.size, np.int32); qe = np.zeros(qid.size, np.int32)
    qs[50:50 + m] = rng.integers(0, 20000, m); qe[50:50 + m] = qs[50:50 + m] + rng.integers(20000, 30000, m)
    qs[:50] = rng.integers(0, 10000, 50); qe[:50] = qs[:50] + 5000
    qs[50 + m:] = 100; qe[50 + m:] = 7000
    p = RaftParams(est_cov=30, symmetric_mode=1)
    want = oracle_run(p, rl, qid, qs, qe, qid, qs, qe)      # (self overlaps: the query sides are all there is, whatever the flag)
    want["symmetric"] = 1
    assert want["cov"].max() >= 32768
    eng = engine.Engine(p, device=0)
    for _ in range(2):
        got, s = engine_result(eng, (rl, qid, qs, qe))
        assert_same_result(got, want, "deep tile")
    eng.close()


def test_a_deep_tile_that_overflows_the_exception_list():
    """A deep pile in one byte per window lists more windows at or above 255 than the default list holds (max(4096, B / 64)): the
    pass is run again with room, deep tiles and all (ADVICE r04 was about this pair of re-runs on the int32 kernels)."""
    from test_gpu_packed_output import check_against, run_width
    rng = np.random.default_rng(11)
    rl = np.array([30000, 12000, 400000, 8000], np.int32)          # read 2: 8000 windows, most of them 40,000 deep
    m = 40000
    qid = np.concatenate([np.zeros(50, np.int32), np.full(m, 2, np.int32), np.full(30, 3, np.int32)])
    qs = np.zeros(qid.size, np.int32); qe = np.zeros(qid.size, np.int32)
    qs[50:50 + m] = rng.integers(0, 20000, m); qe[50:50 + m] = qs[50:50 + m] + rng.integers(330000, 380000, m)
    qs[:50] = rng.integers(0, 10000, 50); qe[:50] = qs[:50] + 5000
    qs[50 + m:] = 100; qe[50 + m:] = 7000
    p = RaftParams(est_cov=30, symmetric_mode=1)
    cols = (rl, qid, qs, qe, qid, qs, qe)
    want = oracle_run(p, *cols)
    want["symmetric"] = 1
    assert want["cov"].max() >= 32768 and int((want["cov"] >= 255).sum()) > 4096
    for width in (1, 2):
        res = run_width(p, cols, width)
        check_against(res, want, width, f"deep tile, width {width}")


def test_cut_points_in_the_pass_or_on_demand():
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    o = make_overlaps(4000, seed=77)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30)
    want = oracle_run(p, *cols)
    for on in (True, False):
        eng = engine.Engine(p, device=0)
        eng.set_emit_cuts(on)
        got, s = engine_result(eng, cols)
        assert_same_result(got, want, f"emit_cuts={on}")
        assert s.n_cuts == want["cuts"].size
        eng.close()


def test_pipeline_derives_offsets_and_window_records_from_the_columns():
    from raft_amd import engine, hostio
    from raft_amd.synth import make_overlaps
    o = make_overlaps(9000, seed=41)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=30, symmetric_mode=1)
    want = oracle_run(RaftParams(est_cov=30), *cols)
    eng = engine.Engine(p, device=0)
    other = engine.Engine(p, device=0)

    def check(res, s, what):
        if "cov_nib" in res:
            cov = hostio.unpack_coverage_d4(s.n_bins, res["cov_nib"], res["cov_anchor"], res["exc_index"], res["exc_value"])
        else:
            cov = hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"])
        assert np.array_equal(cov, want["cov"]), what
        for k in ("cov_offset", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_begin", "frag_end"):
            assert np.array_equal(res[k], want[k]), (what, k)
        assert (s.n_fragments, s.total_coverage, s.total_repeat_length) == (want["frag_begin"].size, want["total_coverage"], want["total_repeat_length"])
    for width in (1, 8):
        for n_chunks, others in ((3, None), (7, None), (5, [other]), (23, [other])):
            out = eng.host_output_buffers(cols[0], pinned=False, width=width)
            res, s = eng.run_pipelined(cols[0], cols[1], cols[2], cols[3], n_chunks=n_chunks, out=out, others=others)
            check(res, s, f"width {width} chunks {n_chunks} ctx {1 + len(others or [])}")
    # what the derivation gives up on ends as the one-piece pass ends
    c2 = [c.copy() for c in cols]
    for c in c2[1:]:
        c[[100, 5000]] = c[[5000, 100]]                                            # two records out of place: still the right result
    res, s = eng.run_pipelined(c2[0], c2[1], c2[2], c2[3], n_chunks=4)
    one = engine.Engine(p, device=0)
    one.run_host(c2[0], c2[1], c2[2], c2[3]); s1 = one.finish(); g1 = one.fetch()
    assert np.array_equal(g1["cov"], want["cov"])
    assert np.array_equal(hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"]), g1["cov"]) and s.n_fragments == s1.n_fragments
    b2 = cols[2].copy(); b2[777] = -5                                              # a negative coordinate: the plain entries' code and index
    with pytest.raises(engine.RaftError) as e:
        eng.run_pipelined(cols[0], cols[1], b2, cols[3], n_chunks=4)
    assert e.value.code == engine.ERR_COORD and e.value.index == 777
    rl2 = cols[0].copy(); rl2[5] = 65_536 * 50 + 10                                # a read of more windows than a window record's 16 bits hold
    res, s = eng.run_pipelined(rl2, cols[1], cols[2], cols[3], n_chunks=4)
    one.run_host(rl2, cols[1], cols[2], cols[3]); s1 = one.finish(); g1 = one.fetch()
    assert np.array_equal(hostio.unpack_coverage(res["cov8"], res["exc_index"], res["exc_value"]), g1["cov"]) and s.n_bins == s1.n_bins
    for e_ in (eng, other, one):
        e_.close()


def test_input_columns_in_memory_from_the_engine_give_the_same_result():
    """raft_hip_device_alloc (ABI 8): device memory placed as the engine places its own large arrays (a virtual range over
    physical chunks spread over the device) -- columns copied there are plain device input; free and reuse work; a buffer
    below the mapping's threshold comes from hipMalloc and behaves the same."""
    import torch
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    o = make_overlaps(60_000, seed=91, device="cuda:0")            # (17 M intervals: the columns are above the 64 MiB threshold)
    p = RaftParams(est_cov=30)
    cols = (o.read_len,) + o.columns()
    eng = engine.Engine(p, device=0)
    eng.run_device(*cols)
    s0 = eng.finish()
    ref = {k: v.clone() for k, v in eng.outputs_device().items()}
    placed = tuple(eng.device_copy(c) for c in cols)
    assert all(torch.equal(a, b) for a, b in zip(placed, cols))
    small = eng.device_tensor(1000, torch.int32)                    # (hipMalloc path)
    small.fill_(7)
    assert int(small.sum()) == 7000
    eng.run_device(*placed)
    s1 = eng.finish()
    out = eng.outputs_device()
    assert (s1.n_fragments, s1.n_repeats, s1.total_coverage) == (s0.n_fragments, s0.n_repeats, s0.total_coverage)
    for k in ref:
        assert torch.equal(out[k], ref[k]), k
    eng.device_free(placed[1])
    again = eng.device_copy(cols[1])                               # (freed chunks are handed out again)
    assert torch.equal(again, cols[1])
    with pytest.raises(engine.RaftError):
        eng.device_free(again[10:])                                # (not a pointer this context handed out)
    eng.close()


def test_placement_pool_is_bounded_and_goes_back_to_the_driver():
    """VERDICT r04 item 7 / ADVICE: the spare chunks of the placement (DevBuf, engine.hip) wait in a per-device pool; the pool
    is capped (RAFT_VMM_POOL_GB), raft_hip_trim hands it back on demand, and the device's last context to be destroyed hands
    all of it back: free device memory returns to within 1 GB of where it was.  Own process: the count of live contexts is exact."""
    import subprocess, sys, os
    code = r'''
import numpy as np, torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
o = make_overlaps(500000, mean_len=30000.0, coverage=8.0, seed=3, device="cuda:0")     # 3e8 windows: a coverage array of 1.2 GB is spread
cols = (o.read_len,) + o.columns()
torch.cuda.synchronize(); torch.cuda.empty_cache()
free0 = torch.cuda.mem_get_info(0)[0]
eng = engine.Engine(RaftParams(est_cov=8))
eng.run_device(*cols); s = eng.finish()
assert s.total_windows * 4 >= (1 << 30)
pooled = engine.pool_bytes(0)
assert 0 < pooled <= 2 * (1 << 30), pooled          # (RAFT_VMM_POOL_GB=2 in this process: the cap holds)
got = engine.trim(0, 1 << 30)
assert engine.pool_bytes(0) <= (1 << 30) and got == pooled - engine.pool_bytes(0)
eng2 = engine.Engine(RaftParams(est_cov=8))
eng.close()
assert engine.pool_bytes(0) > 0                     # (another context lives: the released buffers' chunks wait for it)
eng2.close()
assert engine.pool_bytes(0) == 0
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info(0)[0]
assert free0 - free1 < (1 << 30), (free0, free1)
print("POOL_OK")
'''
    env = dict(os.environ, RAFT_VMM_POOL_GB="2", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("RAFT_NO_VMM", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "POOL_OK" in r.stdout, r.stdout[-2000:]


def test_placement_trial_of_the_coverage_array_keeps_results_and_reports():
    """Round 5: a context whose first pass makes an int32 coverage array of 1 GiB or more runs the pileup kernel into a few candidate
    arrays in that pass and keeps the fastest.  Round 6: only a context that asked for it (raft_hip_set_placement_trial) -- a context
    that did not runs no trial.  Own process (the trial is off once the policy was set by hand anywhere in a process): results of the
    trial pass and of the next one are the same and sane, the report is there."""
    import subprocess, sys, os
    code = r'''
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
o = make_overlaps(500000, mean_len=30000.0, coverage=8.0, seed=3, device="cuda:0")
cols = (o.read_len,) + o.columns()
eng0 = engine.Engine(RaftParams(est_cov=8))          # the default: no trial
eng0.run_device(*cols); s0 = eng0.finish()
assert eng0.placement_trial() is None
eng0.close()
eng = engine.Engine(RaftParams(est_cov=8))
eng.set_placement_trial(4)
sig = []
for rep in range(3):
    eng.run_device(*cols); s = eng.finish()
    out = eng.outputs_device()
    sig.append((s.n_fragments, s.n_repeats, s.total_coverage, s.total_repeat_length, int(out["cov"].sum(dtype=torch.int64)), int(out["rep_s"].sum(dtype=torch.int64))))
    if rep == 0:
        tr = eng.placement_trial()
        assert tr is not None and tr[0] > 0 and tr[1] > 0, tr
assert sig[0] == sig[1] == sig[2] and sig[0][2] == sig[0][4] and s0.n_fragments == sig[0][0], sig
eng2 = engine.Engine(RaftParams(est_cov=8))
eng2.set_placement_trial(4)
engine.set_placement(8)                      # chosen by hand: no trial from here on
eng2.run_device(*cols); s2 = eng2.finish()
assert eng2.placement_trial() is None and s2.n_fragments == sig[0][0]
print("TRIAL_OK", tr)
'''
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for k in ("RAFT_NO_VMM", "RAFT_VMM_SPREAD", "RAFT_PLACEMENT_TRIALS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "TRIAL_OK" in r.stdout, r.stdout[-2000:]
