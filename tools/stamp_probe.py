#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the pileup kernel (variant 3 = variant 0 + stamps).

Stamps per tile (thread 0): 0 kernel entry, 1 after tile descriptor loads, 2 after LDS clear + offset
table, 3 after interval phase, 4 after pass A, 5 own wave done with pass B, 6 all waves done, 7 exit.
Read SHARES, not lengths (the stamps serialise scalar memory).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
o = make_overlaps(reads, mean_len=30000.0, coverage=32.0, seed=7, device="cuda:0")
eng = engine.Engine(RaftParams(est_cov=32))
eng.set_tuning(0, False, 3)
for _ in range(2):
    eng.run_device(o.read_len, *o.columns()); s = eng.finish()
st = eng.debug_stamps().astype(np.int64)
pile, tot = eng.timing()
ok = st[:, 7] > 0
st = st[ok]
print(f"tiles {len(st)}  kernel {pile*1e3:.3f} ms")
d = np.diff(st[:, :8], axis=1)
names = ["descr unpack+prefetch issue", "clear+offset table", "interval phase", "pass A", "wait loads + pass B (own wave)", "wait other waves", "stitch"]
life = st[:, 7] - st[:, 0]
print(f"lifetime cycles: median {np.median(life):.0f} mean {life.mean():.0f} p90 {np.percentile(life,90):.0f}")
for i, n in enumerate(names):
    print(f"  {n:22s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  share {d[:, i].sum()/life.sum():.3f}")
w = st[:, 15] - st[:, 4]
print(f"  4->15 vmcnt(0) before pass B  median {np.median(w):8.0f} mean {w.mean():8.0f}")
sub = np.stack([st[:, 11] - st[:, 6], st[:, 13] - st[:, 11], st[:, 14] - st[:, 13], st[:, 7] - st[:, 14]], 1)
for n, c in zip(["  6->11 resolve seams", "  11->13 emit parked runs", "  13->14 barrier", "  14->7 publish counts"], sub.T):
    print(f"{n:28s} median {np.median(c):8.0f} mean {c.mean():8.0f}")
rt = (st[:, 10] - st[:, 9])
print("memtime ticks per 100MHz realtime tick:", np.median(life[rt > 0] / rt[rt > 0]))
# concurrency: kernel span vs sum of lifetimes
span = st[:, 7].max() - st[:, 0].min()
print(f"span {span} ticks; sum lifetimes/span = {life.sum()/span:.1f} tiles in flight on average")
