#!/usr/bin/env python3
"""A/B of pileup kernel VARIANTS inside ONE process and ONE context (same buffers, same physical placement).
usage: variant_probe.py v1,v2,... [reps]   env: PROBE_FORM=columns|windows PROBE_WIDTH=4|1|2|8 PROBE_READS=n PROBE_WORKLOAD=hg002|ultralong"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raft_amd import engine, hostio
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
vals = [int(v) for v in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
width = int(os.environ.get("PROBE_WIDTH", "4"))
form = os.environ.get("PROBE_FORM", "columns")
wl = os.environ.get("PROBE_WORKLOAD", "hg002")
if wl == "ultralong":
    o = make_overlaps(int(os.environ.get("PROBE_READS", "400000")), mean_len=150000.0, coverage=60.0, seed=20241008, device="cuda:0")
else:
    o = make_overlaps(int(os.environ.get("PROBE_READS", "3300000")), mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
p = RaftParams(est_cov=32 if wl == "hg002" else 60, symmetric_mode=1 if form == "windows" else -1)
eng = engine.Engine(p)
if width != 4:
    eng.set_output_width(width)
if form == "windows":
    off = eng.device_copy(torch.as_tensor(hostio.group_offsets(o.n_reads, o.qid.cpu().numpy())).cuda())
    win = eng.device_copy(torch.as_tensor(hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), 50).view("int32")).cuda())
    rl = eng.device_copy(o.read_len)
    n_bins = int(((o.read_len.long() + 49) // 50).sum())
    run = lambda: eng.run_device_windows(rl, off, win, n_bins=n_bins)
else:
    cols = tuple(eng.device_copy(c) for c in (o.read_len,) + o.columns())
    run = lambda: eng.run_device(*cols)
res = {v: [] for v in vals}
sig = {}
for v in vals:
    eng.set_tuning(0, False, v)
    for _ in range(3):
        run(); s = eng.finish()
    sig[v] = (s.n_fragments, s.total_coverage, s.n_repeats, s.total_repeat_length)
for r in range(reps):
    for v in vals:
        eng.set_tuning(0, False, v)
        k = pp = 0.0
        for _ in range(10):
            run(); s = eng.finish(); a, b = eng.timing(); k += a; pp += b
        res[v].append((k * 100, pp * 100))
print(f"# {wl} {form} width {width}")
for v in vals:
    print(f"variant={v}: kernel " + " ".join(f"{a:.3f}" for a, _ in res[v]) + "   pass " + " ".join(f"{b:.3f}" for _, b in res[v]), "sig", sig[v])
assert len(set(sig.values())) == 1, "variants disagree"
