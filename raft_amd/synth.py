"""Synthetic all-vs-all overlap sets in the reference's PAF column layout.

The reference ships no data; BASELINE.json's configs are restated
here as seeded generators.  What is produced is what the path
sees after PAF tokenisation and name->id resolution (chop.hpp:157-163): int32
columns ``qid, qs, qe, tid, ts, te`` plus ``read_len``.  Sequence bases never
matter to the path (only lengths do), so none are generated.

Model: reads with log-normal lengths are placed uniformly on a genome of length
sum(len)/coverage; every pair of reads sharing >= ``min_ovl`` bp yields an
overlap record (both directions when ``symmetric``, as hifiasm writes them).
Each read carries a haplotype bit; same-haplotype pairs form the "cis" file and
the others the "trans" file, each grouped by ascending query id, and the two are
concatenated -- the shape of ``cat getOverlaps.0.ovlp.paf getOverlaps.1.ovlp.paf``
in the reference's workflow (README.md:36-38).  ``n_families`` repeat families
with ``copies`` copies each add overlaps between reads that cover different
copies, restricted to the repeat, so that coverage inside a repeat exceeds
``est_cov * cov_mul`` and long_repeats is non-empty.

All tensor ops are torch ops so that the 10^8-record set of config 3 can be
generated on the GPU in about a second; on CPU the same code serves the tests.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


@dataclass
class OverlapSet:
    read_len: torch.Tensor  # int32 [N]
    qid: torch.Tensor       # int32 [n_rec]
    qs: torch.Tensor
    qe: torch.Tensor
    tid: torch.Tensor
    ts: torch.Tensor
    te: torch.Tensor
    n_cis: int = 0          # records [0, n_cis) are the cis file, the rest the trans file

    @property
    def n_reads(self) -> int:
        return int(self.read_len.numel())

    @property
    def n_rec(self) -> int:
        return int(self.qid.numel())

    def columns(self):
        return (self.qid, self.qs, self.qe, self.tid, self.ts, self.te)

    def to(self, device) -> "OverlapSet":
        return OverlapSet(self.read_len.to(device), *[c.to(device) for c in self.columns()], n_cis=self.n_cis)

    def take_reads(self, n: int) -> "OverlapSet":
        """Records whose query AND target are among the first ``n`` reads (a closed sub-problem)."""
        keep = (self.qid < n) & (self.tid < n)
        n_cis = int(keep[: self.n_cis].sum())
        return OverlapSet(self.read_len[:n].clone(), *[c[keep] for c in self.columns()], n_cis=n_cis)


def query_window(o: OverlapSet, a: int, b: int) -> OverlapSet:
    """Reads [a, b) of a SYMMETRIC set as a closed problem (a bounded sample of the same workload for CPU checkers).

    In symmetric mode a read's outputs depend only on the records whose QUERY is that read (repeat.hpp:48-58), so the
    window keeps every record with a query in [a, b), rebases the read ids, and maps targets outside the window to one
    extra dummy read (index b - a, as long as the longest read).  Record 0's mirror is appended on the dummy read when
    it is not already inside the window, so that the reference's detection (chop.hpp:175-184) still arrives at
    symmetric = 1.  Outputs of reads 0 .. b-a-1 of the window equal those of reads a .. b-1 of the full set."""
    sel = (o.qid >= a) & (o.qid < b)
    q, qs, qe, t, ts, te = (c[sel] for c in o.columns())
    n = b - a
    inside = (t >= a) & (t < b)
    rl = torch.cat([o.read_len[a:b], o.read_len.max().reshape(1)])
    q = q - a
    t = torch.where(inside, t - a, torch.full_like(t, n))
    cols = [q, qs, qe, t, ts, te]
    if q.numel() and int(t[0]) == n:                      # record 0's target is outside: plant its mirror on the dummy
        extra = [t[:1], ts[:1], te[:1], q[:1], qs[:1], qe[:1]]
        cols = [torch.cat([c, x]) for c, x in zip(cols, extra)]
    return OverlapSet(rl.contiguous(), *[c.contiguous() for c in cols], n_cis=0)


def _expand_ranges(lo: torch.Tensor, cnt: torch.Tensor):
    """For every row i emit (i, lo[i] + k) for k in [0, cnt[i])."""
    total = int(cnt.sum())
    rows = torch.repeat_interleave(torch.arange(lo.numel(), device=lo.device), cnt, output_size=total)
    excl = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(total, device=lo.device) - excl[rows]
    return rows, lo[rows] + k


def _local(strand, length, a, b):
    """Genome-relative offsets [a, b) of a read -> read-local PAF coordinates (reverse reads flip)."""
    fa = torch.where(strand, length - b, a)
    fb = torch.where(strand, length - a, b)
    return fa, fb


def make_overlaps(n_reads: int, mean_len: float = 20000.0, coverage: float = 30.0, seed: int = 1,
                  device: str | torch.device = "cpu", sigma: float = 0.5, min_len: int = 2000,
                  max_len: int = 200000, min_ovl: int = 500, symmetric: bool = True,
                  n_families: int | None = None, copies: int = 2,
                  rep_len: tuple[int, int] = (15000, 50000), group_by_query: bool = True,
                  shuffle: bool = False) -> OverlapSet:
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    i64 = torch.int64

    mu = math.log(mean_len) - 0.5 * sigma * sigma
    lens = torch.exp(mu + sigma * torch.randn(n_reads, generator=g, device=dev, dtype=torch.float64))
    lens = lens.clamp(min_len, max_len).to(i64)
    G = max(int(lens.sum().item() / coverage), int(lens.max().item()) + 1)
    start = (torch.rand(n_reads, generator=g, device=dev, dtype=torch.float64) * (G - lens).to(torch.float64)).to(i64)
    hap = torch.rand(n_reads, generator=g, device=dev) < 0.5
    strand = torch.rand(n_reads, generator=g, device=dev) < 0.5

    order = torch.argsort(start)
    S = start[order]
    E = S + lens[order]

    # positional overlaps: for sorted read i, partners j > i with S[j] < E[i] - min_ovl
    hi = torch.searchsorted(S, E - min_ovl, right=False)
    idx = torch.arange(n_reads, device=dev)
    cnt = (hi - (idx + 1)).clamp(min=0)
    ii, jj = _expand_ranges(idx + 1, cnt)
    a_s, a_e = S[jj], torch.minimum(E[ii], E[jj])          # shared genome segment
    ri, rj = order[ii], order[jj]                             # read ids
    pieces = [(ri, a_s - S[ii], a_e - S[ii], rj, a_s - S[jj], a_e - S[jj])]

    # repeat families
    if n_families is None:
        n_families = max(n_reads // 250, 1)
    if n_families > 0 and copies >= 2:
        nc = n_families * copies
        rl = torch.randint(rep_len[0], rep_len[1] + 1, (n_families,), generator=g, device=dev, dtype=i64)
        rl = rl.clamp(max=max(G // 4, min_ovl + 1))
        rlc = rl.repeat_interleave(copies)
        fam = torch.arange(n_families, device=dev).repeat_interleave(copies)
        cpy = torch.arange(copies, device=dev).repeat(n_families)
        p = (torch.rand(nc, generator=g, device=dev, dtype=torch.float64) * (G - rlc).to(torch.float64)).to(i64)
        # reads sharing >= min_ovl with [p, p+rl): S < p+rl-min_ovl and E > p+min_ovl
        lo = torch.searchsorted(S, p - max_len, right=False)
        hi2 = torch.searchsorted(S, p + rlc - min_ovl, right=False)
        cidx, h_read = _expand_ranges(lo, (hi2 - lo).clamp(min=0))
        ok = E[h_read] > p[cidx] + min_ovl
        cidx, h_read = cidx[ok], h_read[ok]
        u = torch.maximum(S[h_read], p[cidx]) - p[cidx]
        v = torch.minimum(E[h_read], p[cidx] + rlc[cidx]) - p[cidx]
        ok = v - u >= min_ovl
        cidx, h_read, u, v = cidx[ok], h_read[ok], u[ok], v[ok]
        BIG = int(rep_len[1]) * 4 + 8
        key = fam[cidx] * BIG + u
        ko = torch.argsort(key)
        key, cidx, h_read, u, v = key[ko], cidx[ko], h_read[ko], u[ko], v[ko]
        tgt = fam[cidx] * BIG + (v - min_ovl)
        hh = torch.searchsorted(key, tgt, right=False)
        hidx = torch.arange(key.numel(), device=dev)
        a, b = _expand_ranges(hidx + 1, (hh - (hidx + 1)).clamp(min=0))
        ok = (cpy[cidx[a]] != cpy[cidx[b]]) & (h_read[a] != h_read[b])
        a, b = a[ok], b[ok]
        ru, rv = u[b], torch.minimum(v[a], v[b])            # repeat-local shared segment
        ok = rv - ru >= min_ovl
        a, b, ru, rv = a[ok], b[ok], ru[ok], rv[ok]
        ga, gb = p[cidx[a]], p[cidx[b]]
        ra, rb = h_read[a], h_read[b]
        pieces.append((order[ra], ga + ru - S[ra], ga + rv - S[ra], order[rb], gb + ru - S[rb], gb + rv - S[rb]))

    A = torch.cat([x[0] for x in pieces]); As = torch.cat([x[1] for x in pieces]); Ae = torch.cat([x[2] for x in pieces])
    Bq = torch.cat([x[3] for x in pieces]); Bs = torch.cat([x[4] for x in pieces]); Be = torch.cat([x[5] for x in pieces])
    As, Ae = _local(strand[A], lens[A], As, Ae)
    Bs, Be = _local(strand[Bq], lens[Bq], Bs, Be)

    if symmetric:
        qid = torch.cat([A, Bq]); tid = torch.cat([Bq, A])
        qs = torch.cat([As, Bs]); qe = torch.cat([Ae, Be])
        ts = torch.cat([Bs, As]); te = torch.cat([Be, Ae])
    else:
        flip = torch.rand(A.numel(), generator=g, device=dev) < 0.5
        qid = torch.where(flip, Bq, A); tid = torch.where(flip, A, Bq)
        qs = torch.where(flip, Bs, As); qe = torch.where(flip, Be, Ae)
        ts = torch.where(flip, As, Bs); te = torch.where(flip, Ae, Be)

    trans = hap[qid] != hap[tid]
    n_cis = int((~trans).sum())
    if shuffle:
        perm = torch.randperm(qid.numel(), generator=g, device=dev)
        n_cis = 0
    elif group_by_query:
        # cis file then trans file, each ascending in (query, target); N < 2^30 keeps the key in int64
        perm = torch.argsort(trans.to(i64) * (n_reads * n_reads) + qid * n_reads + tid, stable=True)
    else:
        perm = torch.arange(qid.numel(), device=dev)
        n_cis = 0
    i32 = torch.int32
    cols = [c[perm].to(i32).contiguous() for c in (qid, qs, qe, tid, ts, te)]
    return OverlapSet(lens.to(i32).contiguous(), *cols, n_cis=n_cis)
