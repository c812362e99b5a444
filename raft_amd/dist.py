"""Sharding of the RAFT hot path across the GPUs of one node (one process per GPU, torch.distributed).

The path shards by READS: a read's coverage, repeats and cut points depend only on the intervals that
land on that read (repeat.hpp:99-171, chop.hpp:198-246).  Rank g owns a contiguous FASTA-index range
[bounds[g], bounds[g+1]); outputs concatenate in rank order, so file order is unchanged.  Cross-read
state is tiny: the symmetric flag (a property of the whole PAF, chop.hpp:175-184), the global fragment
counter read_num (chop.hpp:195) and the four stdout sums (repeat.hpp:93-97).

Two ways to get each rank its intervals:
  * host-routed (default, no collective): whoever tokenises the PAF knows the owner of every interval.
  * pre-split PAF (BASELINE configs[3]): rank g holds an arbitrary contiguous slice of the records.  It
    expands them to intervals, buckets them by owner and ONE exchange step -- an all-to-all-v per int32 column (RCCL over xGMI
    when the tensors are on GPUs, gloo on CPU) -- delivers them.  xGMI is point-to-point, every ordered pair has its
    own link, so the exchange is a single step of n*(n-1) independent transfers.

Everything here is tensor plumbing (torch ops + torch.distributed); the per-rank compute is the HIP
engine (raft_amd.engine.Engine) fed with intervals as query-side records, symmetric_mode = 1.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


def partition_reads(read_len: torch.Tensor, reso: int, n_parts: int, intervals_per_read: torch.Tensor | None = None,
                    interval_weight: float = 3.0) -> torch.Tensor:
    """Contiguous read ranges of near-equal weight  w_i = windows_i + interval_weight * intervals_i.

    Returns int64 bounds[n_parts + 1] with bounds[0] = 0 and bounds[-1] = n_reads.  12 B are read per
    interval and 4 B written per window, hence the default weight of 3 windows per interval.
    """
    n = int(read_len.numel())
    w = (read_len.to(torch.int64) + (reso - 1)) // reso
    if intervals_per_read is not None:
        w = w + (interval_weight * intervals_per_read.to(torch.float64)).to(torch.int64)
    cum = torch.cumsum(w, 0)
    total = int(cum[-1]) if n else 0
    targets = torch.tensor([total * g // n_parts for g in range(1, n_parts)], dtype=torch.int64, device=read_len.device)
    if n and n_parts > 1:
        inner = torch.searchsorted(cum, targets, right=False) + 1
    else:
        inner = torch.zeros(max(n_parts - 1, 0), dtype=torch.int64)
    inner = inner.clamp(max=n).to(torch.int64).cpu()
    bounds = torch.cat([torch.zeros(1, dtype=torch.int64), inner, torch.tensor([n], dtype=torch.int64)])
    return torch.cummax(bounds, 0).values


def detect_symmetric(cols, first_record=None) -> bool:
    """chop.hpp:171-184 on a slice: does any record (other than global record 0) mirror record 0?

    `first_record` = the six integers of global record 0 (None: the slice starts with it)."""
    qid, qs, qe, tid, ts, te = cols
    if qid.numel() == 0:
        return False
    if first_record is None:
        f = [int(c[0]) for c in cols]
        body = [c[1:] for c in cols]
    else:
        f = [int(x) for x in first_record]
        body = cols
    q0, qs0, qe0, t0, ts0, te0 = f
    m = (body[0] == t0) & (body[3] == q0) & (body[4] == qs0) & (body[5] == qe0) & (body[1] == ts0) & (body[2] == te0)
    return bool(m.any())


def expand_intervals(cols, symmetric: bool):
    """Records -> the multiset of (read, start, end) intervals the reference piles up (repeat.hpp:48-58)."""
    qid, qs, qe, tid, ts, te = cols
    if symmetric:
        return qid, qs, qe
    keep = tid != qid
    return torch.cat([qid, tid[keep]]), torch.cat([qs, ts[keep]]), torch.cat([qe, te[keep]])


def route_intervals_host(cols, bounds: torch.Tensor, symmetric: bool):
    """Host-routed mode: per-rank (local_read, start, end) lists from the full record set (no collective)."""
    rid, s, e = expand_intervals(cols, symmetric)
    owner = torch.searchsorted(bounds[1:].to(rid.device), rid.to(torch.int64), right=True)
    out = []
    for g in range(bounds.numel() - 1):
        m = owner == g
        out.append(((rid[m] - int(bounds[g])).to(torch.int32), s[m], e[m]))
    return out


def exchange_intervals(cols_local, bounds: torch.Tensor, symmetric: bool, group=None):
    """Pre-split mode: one exchange step (all-to-all-v of the three columns) routes this rank's intervals to their owners.

    Returns (local_read, start, end) of the intervals this rank owns, as int32 tensors on the input device."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    rid, s, e = expand_intervals(cols_local, symmetric)
    dev = rid.device
    owner = torch.searchsorted(bounds[1:].to(dev), rid.to(torch.int64), right=True)
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    n_recv = int(sum(rc))
    out = []
    for col in (rid, s, e):                              # SoA stays SoA: one all-to-all-v per int32 column
        got = torch.empty(n_recv, dtype=torch.int32, device=dev)
        dist.all_to_all_single(got, col[order].contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=group)
        out.append(got)
    out[0] -= int(bounds[rank])                          # read ids local to this rank's range
    return tuple(out)


def global_symmetric_flag(cols_local, group=None) -> bool:
    """Pre-split mode: record 0 lives on rank 0; every rank tests its slice against it; OR over ranks."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = cols_local[0].device
    first = torch.zeros(7, dtype=torch.int64, device=dev)
    if rank == 0 and cols_local[0].numel() > 0:
        first[:6] = torch.stack([c[0].to(torch.int64) for c in cols_local])
        first[6] = 1
    dist.broadcast(first, src=0, group=group)
    if int(first[6]) == 0:
        return False
    f = first[:6].tolist()
    mine = detect_symmetric(cols_local, None if rank == 0 else f) if cols_local[0].numel() else False
    flag = torch.tensor([1 if mine else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(int(flag))


@dataclass
class ShardTotals:
    """What has to cross ranks after the local passes: fragment id bases and the stdout sums."""
    frag_base: int                # read_num - 1 of this rank's first fragment (chop.hpp:195)
    n_fragments: int
    total_coverage: int
    total_windows: int
    total_repeat_length: int
    total_read_length: int


def combine_totals(n_fragments: int, total_coverage: int, total_windows: int, total_repeat_length: int,
                   total_read_length: int, device="cpu", group=None) -> ShardTotals:
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([n_fragments, total_coverage, total_windows, total_repeat_length, total_read_length],
                        dtype=torch.int64, device=device)
    allv = torch.empty((world, 5), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allv, mine.unsqueeze(0), group=group)
    allv = allv.cpu()
    tot = allv.sum(0).tolist()
    return ShardTotals(frag_base=int(allv[:rank, 0].sum()), n_fragments=tot[0], total_coverage=tot[1], total_windows=tot[2],
                       total_repeat_length=tot[3], total_read_length=tot[4])


def run_shard(engine, read_len_local: torch.Tensor, intervals):
    """Feeds this rank's intervals to the HIP engine as query-side records (symmetric_mode must be 1)."""
    rid, s, e = intervals
    engine.run_device(read_len_local, rid, s, e, rid, s, e)
    return engine.finish()
