"""GPU, multi-process: the sharding layer (raft_amd/dist.py) driving the HIP ENGINE on every rank.

World sizes 2 and 3 over gloo with all ranks sharing the box's one MI355X (the same code path `bench.py --gpus N` and
the 8-GPU run take, minus RCCL).  Each rank partitions the reads, gets its intervals either host-routed or through the
pre-split all-to-all-v exchange, runs `run_shard` on a real Engine with symmetric_mode = 1 (the routed intervals are
unsorted, so the engine takes its counting-sort path), and the concatenation of the shards -- coverage, repeats,
fragments with the global read_num base, the stdout sums -- must equal a single-engine pass and the oracle.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
from raft_testlib import RaftParams, assert_same_result, oracle_run

pytestmark = pytest.mark.gpu

KEYS = ("cov", "rep_s", "rep_e", "cuts", "frag_begin", "frag_end")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kw, params, mode, outdir):
    import torch
    import torch.distributed as dist

    from raft_amd import dist as rdist
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = make_overlaps(**kw)
        p = RaftParams(**params)
        per_read = torch.bincount(o.qid.long(), minlength=o.n_reads)
        bounds = rdist.partition_reads(o.read_len, p.reso, world, per_read)
        b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
        if mode == "presplit":
            lo, hi = o.n_rec * rank // world, o.n_rec * (rank + 1) // world
            cols = [c[lo:hi] for c in o.columns()]
            sym = rdist.global_symmetric_flag(cols)
            rid, s, e = rdist.exchange_intervals(cols, bounds, sym)
        else:
            sym = rdist.detect_symmetric(o.columns())
            rid, s, e = rdist.route_intervals_host(o.columns(), bounds, sym)[rank]
        eng = engine.Engine(RaftParams(**dict(params, symmetric_mode=1)), device=0)
        rl = o.read_len[b0:b1].contiguous().to("cuda:0")
        summ = rdist.run_shard(eng, rl, tuple(t.contiguous().to("cuda:0") for t in (rid, s, e)))
        got = eng.fetch()
        tot = rdist.combine_totals(summ.n_fragments, summ.total_coverage, summ.total_windows, summ.total_repeat_length,
                                   summ.total_read_length)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), sym=int(sym), b0=b0, b1=b1, frag_base=tot.frag_base,
                 path=summ.interval_path, n_intervals=summ.n_intervals,
                 totals=np.array([tot.n_fragments, tot.total_coverage, tot.total_windows, tot.total_repeat_length,
                                  tot.total_read_length]), **got)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["host_routed", "presplit"])
@pytest.mark.parametrize("world,kw", [(2, dict(n_reads=3000, seed=41)),
                                      (3, dict(n_reads=2000, seed=42, symmetric=False, shuffle=True)),
                                      (3, dict(n_reads=2500, seed=43, shuffle=True))])
def test_shards_on_the_engine_equal_single_pass(world, kw, mode):
    import torch.multiprocessing as mp

    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    params = dict(est_cov=30 if kw.get("symmetric", True) else 15)
    o = make_overlaps(**kw)
    p = RaftParams(**params)
    cols = [c.numpy() for c in (o.read_len,) + o.columns()]
    want = oracle_run(p, *cols)
    eng = engine.Engine(p, device=0)                     # the single-engine pass of the same set
    eng.run_host(*cols)
    s = eng.finish()
    single = eng.fetch()
    single.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                  total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
    eng.close()
    assert_same_result(single, want, "single pass")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), kw, params, mode, d), nprocs=world, join=True)
        parts = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    assert all(int(z["sym"]) == want["symmetric"] for z in parts)
    assert parts[0]["b0"] == 0 and parts[-1]["b1"] == o.n_reads and all(parts[i]["b1"] == parts[i + 1]["b0"] for i in range(world - 1))
    assert sum(int(z["n_intervals"]) for z in parts) == want["n_intervals"]
    for k in KEYS:
        assert np.array_equal(np.concatenate([z[k] for z in parts]), single[k]), k
    assert np.array_equal(np.concatenate([z["frag_read"] + z["b0"] for z in parts]), single["frag_read"])
    for key in ("cov", "rep", "cut", "frag"):            # CSR offsets chain across the shards
        off, base = [], 0
        for z in parts:
            off.append(z[key + "_offset"][:-1] + base)
            base += int(z[key + "_offset"][-1])
        assert np.array_equal(np.concatenate(off + [np.array([base])]), single[key + "_offset"]), key
    base = 0
    for z in parts:                                      # global read_num of each shard's first fragment (chop.hpp:195)
        assert int(z["frag_base"]) == base
        base += len(z["frag_read"])
    assert parts[0]["totals"].tolist() == [len(want["frag_read"]), want["total_coverage"], want["total_windows"],
                                           want["total_repeat_length"], want["total_read_length"]]


# ---- BASELINE configs[3] at its own size and world size: eight ranks, each holding a true eighth of the configs[2] record stream ----

def _worker8(rank, world, port, n_reads, seed, est_cov, win_reads, outdir):
    import torch
    import torch.distributed as dist

    from raft_amd import dist as rdist
    from raft_amd import engine
    from raft_amd.synth import make_overlaps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = RaftParams(est_cov=est_cov)
        cols = rl = bounds = None
        for turn in range(world):                        # (the generator's scratch is several GB: one rank at a time on the shared GPU)
            if turn == rank:
                o = make_overlaps(n_reads, mean_len=30000.0, coverage=32.0, seed=seed, device="cuda:0")
                per_read = torch.bincount(o.qid.long(), minlength=o.n_reads)
                bounds = rdist.partition_reads(o.read_len, p.reso, world, per_read)
                lo, hi = o.n_rec * rank // world, o.n_rec * (rank + 1) // world
                cols = [c[lo:hi].cpu() for c in o.columns()]          # this rank's contiguous slice of the PAF stream
                b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
                rl = o.read_len[b0:b1].clone()
                del o, per_read
                torch.cuda.empty_cache()
            dist.barrier()
        sym = rdist.global_symmetric_flag(cols)                       # broadcast of record 0 + MAX all-reduce
        rid, s, e = rdist.exchange_intervals(cols, bounds, sym)       # ONE exchange step: an all-to-all-v per column
        eng = engine.Engine(RaftParams(est_cov=est_cov, symmetric_mode=1), device=0)
        summ = rdist.run_shard(eng, rl, tuple(t.contiguous().to("cuda:0") for t in (rid, s, e)))
        out = eng.outputs_device()
        w = min(win_reads, int(rl.numel()))
        cut = {k: int(out[k + "_offset"][w]) for k in ("cov", "rep", "cut", "frag")}
        tot = rdist.combine_totals(summ.n_fragments, summ.total_coverage, summ.total_windows, summ.total_repeat_length, summ.total_read_length)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), sym=int(sym), b0=b0, b1=b1, w=w, frag_base=tot.frag_base, n_intervals=summ.n_intervals,
                 n_frag=summ.n_fragments, n_rep=summ.n_repeats, n_bins=summ.n_bins, path=summ.interval_path,
                 totals=np.array([tot.n_fragments, tot.total_coverage, tot.total_windows, tot.total_repeat_length, tot.total_read_length]),
                 cov=out["cov"][:cut["cov"]].cpu().numpy(), rep_s=out["rep_s"][:cut["rep"]].cpu().numpy(), rep_e=out["rep_e"][:cut["rep"]].cpu().numpy(),
                 cuts=out["cuts"][:cut["cut"]].cpu().numpy(), frag_begin=out["frag_begin"][:cut["frag"]].cpu().numpy(),
                 frag_end=out["frag_end"][:cut["frag"]].cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


def test_config4_eight_ranks_each_a_true_eighth_of_the_full_size_stream():
    """BASELINE configs[3] as written, minus the eight GPUs: 8 ranks (gloo, sharing the box's MI355X), the ONE configs[2] set
    (3.3 M reads, 2.9e8 records), every rank holding a contiguous EIGHTH of the record stream; the exchange step routes the
    intervals to their owners; every rank runs the HIP engine on its read range.  The ranks' totals equal the single pass over
    the whole set, the read ranges chain, global read_num bases chain, and the first 20 k reads of every rank's range equal
    the oracle on that window cut out of the full set."""
    import psutil
    import torch
    import torch.multiprocessing as mp

    from raft_amd import engine
    from raft_amd.synth import make_overlaps, query_window
    if psutil.virtual_memory().available < 48 * (1 << 30):
        pytest.skip("needs ~30 GB of host memory with room to spare")
    n_reads, seed, est_cov, world, win = 3_300_000, 20241008, 32, 8, 20_000
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker8, args=(world, _free_port(), n_reads, seed, est_cov, win, d), nprocs=world, join=True)
        parts = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    o = make_overlaps(n_reads, mean_len=30000.0, coverage=32.0, seed=seed, device="cuda:0")
    p = RaftParams(est_cov=est_cov)
    eng = engine.Engine(p, device=0)
    eng.run_device(o.read_len, *o.columns())
    s = eng.finish()
    eng.close()
    assert all(int(z["sym"]) == 1 for z in parts) and s.symmetric == 1
    assert parts[0]["b0"] == 0 and parts[-1]["b1"] == n_reads and all(parts[i]["b1"] == parts[i + 1]["b0"] for i in range(world - 1))
    assert sum(int(z["n_intervals"]) for z in parts) == o.n_rec == s.n_intervals
    assert sum(int(z["n_frag"]) for z in parts) == s.n_fragments and sum(int(z["n_rep"]) for z in parts) == s.n_repeats
    assert sum(int(z["n_bins"]) for z in parts) == s.n_bins
    assert parts[0]["totals"].tolist() == [s.n_fragments, s.total_coverage, s.total_windows, s.total_repeat_length, s.total_read_length]
    base = 0
    for z in parts:                                      # global read_num of each rank's first fragment (chop.hpp:195)
        assert int(z["frag_base"]) == base
        base += int(z["n_frag"])
    # a rank's reads number as many as an eighth of the weight allows: none is far from an eighth of the records
    assert max(int(z["n_intervals"]) for z in parts) < 1.25 * o.n_rec / world
    for r, z in enumerate(parts):                        # the oracle on the first reads of every rank's range
        a, b = int(z["b0"]), int(z["b0"]) + int(z["w"])
        qw = query_window(o, a, b)
        want = oracle_run(p, *[c.cpu().numpy() for c in (qw.read_len,) + qw.columns()])
        n = b - a
        for key, off in (("cov", "cov_offset"), ("rep_s", "rep_offset"), ("rep_e", "rep_offset"), ("cuts", "cut_offset"),
                         ("frag_begin", "frag_offset"), ("frag_end", "frag_offset")):
            assert np.array_equal(z[key], want[key][: want[off][n]]), (r, key)
