#!/usr/bin/env python3
"""Intervals per tile on the benchmark set (sizes the prefetch slots of the fast pileup kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd.synth import make_overlaps
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4608
o = make_overlaps(reads, mean_len=30000.0, coverage=32.0, seed=20242008, device="cuda:0")
win = (o.read_len.long() + 49) // 50
cov_off = torch.cumsum(win, 0) - win
tile = cov_off // Q
qid = o.columns()[0].long()
n_cis = o.n_cis
nt = int(tile.max()) + 1
tot = torch.bincount(tile[qid], minlength=nt).float()
s0 = torch.bincount(tile[qid[:n_cis]], minlength=nt).float()
s1 = torch.bincount(tile[qid[n_cis:]], minlength=nt).float()
print(f"tiles {nt}  intervals/tile mean {tot.mean():.0f} p50 {tot.median():.0f} p90 {tot.quantile(0.9):.0f} p99 {tot.quantile(0.99):.0f}")
print(f"segment 0 mean {s0.mean():.0f}, segment 1 mean {s1.mean():.0f}")
for lim in (512, 640, 768, 896, 1024):
    print(f"  tiles with more than {lim:4d} intervals: {(tot > lim).float().mean():.4f}")
print(f"  tiles with a segment above 512 (today's synchronous path): {((s0 > 512) | (s1 > 512)).float().mean():.4f}")
print(f"  tiles with a segment above 256: {((s0 > 256) | (s1 > 256)).float().mean():.4f};  both at most 256: {((s0 <= 256) & (s1 <= 256)).float().mean():.4f}")
