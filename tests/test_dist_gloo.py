"""CPU, multi-process (gloo, world_size 2 and 3): the sharding layer raft_amd/dist.py.

Each rank holds a contiguous slice of the PAF records (pre-split mode), the ranks agree on the symmetric flag,
partition the reads by weight and route intervals with one all-to-all-v.  The per-rank pass is stood in for by the
CPU oracle (test infrastructure; on GPUs it is the HIP engine) and the concatenation of the shards must equal the
oracle's result on the whole input, including fragment numbering and the stdout sums.
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from raft_testlib import RaftParams, oracle_run

from raft_amd import dist as rdist
from raft_amd.synth import make_overlaps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kw, params, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = make_overlaps(**kw)
        p = RaftParams(**params)
        n_rec = o.n_rec
        lo, hi = n_rec * rank // world, n_rec * (rank + 1) // world      # this rank's slice of the record stream
        cols = [c[lo:hi] for c in o.columns()]
        sym = rdist.global_symmetric_flag(cols)
        bounds = rdist.partition_reads(o.read_len, p.reso, world)
        rid, s, e = rdist.exchange_intervals(cols, bounds, sym)
        b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
        rl = o.read_len[b0:b1]
        res = oracle_run(p, rl.numpy(), rid.numpy(), s.numpy(), e.numpy(), rid.numpy(), s.numpy(), e.numpy())
        tot = rdist.combine_totals(len(res["frag_read"]), res["total_coverage"], res["total_windows"],
                                   res["total_repeat_length"], res["total_read_length"])
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), sym=int(sym), b0=b0, b1=b1, frag_base=tot.frag_base,
                 totals=np.array([tot.n_fragments, tot.total_coverage, tot.total_windows, tot.total_repeat_length,
                                  tot.total_read_length]),
                 **{k: res[k] for k in ("cov", "rep_offset", "rep_s", "rep_e", "frag_offset", "frag_read", "frag_begin", "frag_end")})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kw", [(2, dict(n_reads=400, seed=31)),
                                      (3, dict(n_reads=300, seed=32, symmetric=False, shuffle=True)),
                                      (2, dict(n_reads=300, seed=33, shuffle=True))])
def test_presplit_exchange_matches_single_pass(world, kw):
    params = dict(est_cov=30 if kw.get("symmetric", True) else 15)
    o = make_overlaps(**kw)
    p = RaftParams(**params)
    want = oracle_run(p, *[c.numpy() for c in (o.read_len,) + o.columns()])
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), kw, params, d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    assert all(int(z["sym"]) == want["symmetric"] for z in parts)
    assert parts[0]["b0"] == 0 and parts[-1]["b1"] == o.n_reads and all(parts[i]["b1"] == parts[i + 1]["b0"] for i in range(world - 1))
    for k in ("cov", "rep_s", "rep_e", "frag_begin", "frag_end"):
        assert np.array_equal(np.concatenate([z[k] for z in parts]), want[k]), k
    assert np.array_equal(np.concatenate([z["frag_read"] + z["b0"] for z in parts]), want["frag_read"])
    # global read_num of each shard's first fragment
    base = 0
    for z in parts:
        assert int(z["frag_base"]) == base
        base += len(z["frag_read"])
    tot = parts[0]["totals"].tolist()
    assert tot == [len(want["frag_read"]), want["total_coverage"], want["total_windows"], want["total_repeat_length"],
                   want["total_read_length"]]


def test_partition_and_host_routing():
    o = make_overlaps(500, seed=34)
    p = RaftParams(est_cov=30)
    per_read = torch.bincount(o.qid.long(), minlength=o.n_reads)
    b = rdist.partition_reads(o.read_len, p.reso, 4, per_read)
    assert b[0] == 0 and b[-1] == o.n_reads and bool((b[1:] >= b[:-1]).all())
    w = (o.read_len.long() + p.reso - 1) // p.reso + 3 * per_read
    shares = torch.stack([w[b[i]:b[i + 1]].sum() for i in range(4)]).double()
    assert float(shares.max() / shares.mean()) < 1.15                      # balanced within 15 %
    sym = rdist.detect_symmetric(o.columns())
    shards = rdist.route_intervals_host(o.columns(), b, sym)
    want = oracle_run(p, *[c.numpy() for c in (o.read_len,) + o.columns()])
    cov = []
    for g, (rid, s, e) in enumerate(shards):
        rl = o.read_len[int(b[g]):int(b[g + 1])]
        r = oracle_run(p, rl.numpy(), rid.numpy(), s.numpy(), e.numpy(), rid.numpy(), s.numpy(), e.numpy())
        cov.append(r["cov"])
    assert np.array_equal(np.concatenate(cov), want["cov"])
    assert rdist.partition_reads(o.read_len[:0], 50, 3).tolist() == [0, 0, 0, 0]
