#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the pileup kernel (variant 3 = variant 0 + stamps).

Stamps per tile (thread 0) of pileup_fast_kernel: 0 tile start, 1 after cut unpack + prefetch issue, 2 own wave
done with the interval phase, 3 past barrier A, 4 past pass A and barrier B, 15 loads landed, 5 own wave done with
staging the next tile + pass B, 6 past barrier C, 11 seams resolved, 13 runs emitted, 7 tile end.
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
mode = sys.argv[2] if len(sys.argv) > 2 else "columns"       # columns | windows (window records: pileup_fast.hpp IN = 1)
eng = engine.Engine(RaftParams(est_cov=32, symmetric_mode=1 if mode == "windows" else -1))
eng.set_tuning(0, False, 3)
if mode == "windows":
    from raft_amd import hostio
    off = torch.as_tensor(hostio.group_offsets(reads, o.qid.cpu().numpy())).to("cuda:0")
    win = torch.as_tensor(hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), 50).view(np.int32)).to("cuda:0")
for _ in range(2):
    if mode == "windows":
        eng.run_device_windows(o.read_len, off, win)
    else:
        eng.run_device(o.read_len, *o.columns())
    s = eng.finish()
st = eng.debug_stamps().astype(np.int64)
pile, tot = eng.timing()
ok = (st[:, 7] > 0) & (np.diff(st[:, :8], axis=1) >= 0).all(axis=1)   # tiles the regular instantiation processed in full
st = st[ok]
print(f"tiles {len(st)}  kernel {pile*1e3:.3f} ms")
d = np.diff(st[:, :8], axis=1)
names = ["cut unpack+prefetch issue", "interval phase (own)", "barrier A", "pass A + barrier B", "wait loads+stage next+pass B (own)", "barrier C", "seams+emit"]
life = st[:, 7] - st[:, 0]
print(f"lifetime cycles: median {np.median(life):.0f} mean {life.mean():.0f} p90 {np.percentile(life,90):.0f}")
for i, n in enumerate(names):
    print(f"  {n:22s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f}  share {d[:, i].sum()/life.sum():.3f}")
w = st[:, 15] - st[:, 4]
print(f"  4->15 vmcnt(0) before pass B  median {np.median(w):8.0f} mean {w.mean():8.0f}")
sub = np.stack([st[:, 11] - st[:, 6], st[:, 13] - st[:, 11]], 1)
for n, c in zip(["  6->11 resolve seams", "  11->13 emit parked runs"], sub.T):
    print(f"{n:28s} median {np.median(c):8.0f} mean {c.mean():8.0f}")
print(f"tiles with intervals beyond the prefetched slots: {(st[:, 12] != 0).mean():.3f}")
# per-row cost of the two passes: the stamping wave (wave 0) owns ceil(rows / 4) rows of a tile with stamp 8 windows
rpw = (((st[:, 8] + 2 + 1 + 255) >> 8) + 3) // 4
for nm, col in (("pass A + barrier B", 3), ("stage + pass B", 4)):
    A = np.stack([np.ones(len(rpw)), rpw.astype(np.float64)], 1)
    coef, *_ = np.linalg.lstsq(A, d[:, col].astype(np.float64), rcond=None)
    by = {int(r): float(np.median(d[rpw == r, col])) for r in np.unique(rpw) if (rpw == r).sum() > 200}
    print(f"  {nm}: {coef[0]:.0f} cycles + {coef[1]:.0f} per row of wave 0; medians by rows of wave 0: {by}")
rt = (st[:, 10] - st[:, 9])
print("memtime ticks per 100MHz realtime tick:", np.median(life[rt > 0] / rt[rt > 0]))
# concurrency: kernel span vs sum of lifetimes
span = st[:, 7].max() - st[:, 0].min()
print(f"span {span} ticks; sum lifetimes/span = {life.sum()/span:.1f} tiles in flight on average")
# per-workgroup view (persistent grid)
grid = int(os.environ.get("RAFT_PROBE_GRID", "1024"))   # variant 3: 4 workgroups per CU
full = eng.debug_stamps().astype(np.int64)
kidx = np.nonzero(full[:, 7] > 0)[0]
good = (full[kidx, 0] > 0) & (full[kidx, 7] > full[kidx, 0]) & (full[kidx, 7] - full[kidx, 0] < 10_000_000)
kidx = kidx[good]
wg = full[kidx, 14] % grid          # stamp 14: the workgroup that ran the tile (tiles are handed out dynamically)
t0 = full[kidx, 0]; t7 = full[kidx, 7]
first = np.full(grid, np.iinfo(np.int64).max); last = np.zeros(grid, np.int64); busy = np.zeros(grid, np.int64); cnt = np.zeros(grid, np.int64)
np.minimum.at(first, wg, t0); np.maximum.at(last, wg, t7); np.add.at(busy, wg, t7 - t0); np.add.at(cnt, wg, 1)
okw = cnt > 0
origin = first[okw].min()
print(f"workgroups with tiles {okw.sum()}  start spread (cycles after the first): median {np.median(first[okw]-origin):.0f} p90 {np.percentile(first[okw]-origin,90):.0f} max {(first[okw]-origin).max():.0f}")
print(f"workgroup span median {np.median((last-first)[okw]):.0f}  in-tile share of span median {np.median(busy[okw]/(last-first)[okw]):.3f}  kernel span {last[okw].max()-origin}")
order = np.argsort(kidx)
ks, ws, a0, a7 = kidx[order], wg[order], t0[order], t7[order]
nxt = {}
gaps = []
for k_, w_, s0, s7 in zip(ks[::-1], ws[::-1], a0[::-1], a7[::-1]):
    if w_ in nxt: gaps.append(nxt[w_] - s7)
    nxt[w_] = s0
gaps = np.array(gaps)
print(f"gap between tiles of a workgroup (stamp 7 -> next stamp 0): median {np.median(gaps):.0f} mean {gaps.mean():.0f} p90 {np.percentile(gaps,90):.0f}")
# the same on the 100 MHz s_memrealtime clock (common to all XCDs): when does each workgroup start / stop?
r0 = full[kidx, 9]; r1 = full[kidx, 10]
rfirst = np.full(grid, np.iinfo(np.int64).max); rlast = np.zeros(grid, np.int64)
np.minimum.at(rfirst, wg, r0); np.maximum.at(rlast, wg, r1)
o = rfirst[okw].min()
st_us = (rfirst[okw] - o) / 100.0; en_us = (rlast[okw] - o) / 100.0
print("workgroup start (us after first): p10 %.0f p50 %.0f p75 %.0f p85 %.0f p95 %.0f max %.0f" % tuple(np.percentile(st_us, [10, 50, 75, 85, 95, 100])))
print("workgroup end   (us after first): p10 %.0f p50 %.0f p75 %.0f p85 %.0f p95 %.0f max %.0f" % tuple(np.percentile(en_us, [10, 50, 75, 85, 95, 100])))
late = st_us > 0.25 * en_us.max()
print(f"workgroups starting later than 25% into the kernel: {late.sum()} of {okw.sum()}")
