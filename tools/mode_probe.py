#!/usr/bin/env python3
"""A/B of run-time modes of the wave kernel inside ONE process and ONE context (same buffers, same physical placement: kernel
times of separate processes differ by up to 10 % with where their buffers land).  usage: mode_probe.py ENV=v1,v2,... [reps] [bench-set args]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
name, vals = sys.argv[1].split("=")
vals = vals.split(",")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
width = int(os.environ.get("PROBE_WIDTH", "4"))
o = make_overlaps(int(os.environ.get("PROBE_READS", "3300000")), mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
eng = engine.Engine(RaftParams(est_cov=32))
if width != 4:
    eng.set_output_width(width)
cols = tuple(eng.device_copy(c) for c in (o.read_len,) + o.columns())      # (inputs where the engine would put them: raft_hip_device_alloc)
form = os.environ.get("PROBE_FORM", "columns")      # columns | windows (per-read offsets + one word per record, symmetric flag handed over)
if form == "windows":
    import numpy as np
    from raft_amd import hostio
    eng.close()
    eng = engine.Engine(RaftParams(est_cov=32, symmetric_mode=1))
    if width != 4:
        eng.set_output_width(width)
    off = eng.device_copy(torch.as_tensor(hostio.group_offsets(o.n_reads, o.qid.cpu().numpy())).cuda())
    win = eng.device_copy(torch.as_tensor(hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), 50).view("int32")).cuda())
    rl = eng.device_copy(o.read_len)
    n_bins = int(((o.read_len.long() + 49) // 50).sum())
    run = lambda: eng.run_device_windows(rl, off, win, n_bins=n_bins)
else:
    run = lambda: eng.run_device(*cols)
for _ in range(5):
    run(); eng.finish()
res = {v: [] for v in vals}
for r in range(reps):
    for v in vals:
        if name == "Q":
            eng.set_tuning(int(v), False, -1)
        else:
            os.environ[name] = v
        k = p = 0.0
        for _ in range(10):
            run(); s = eng.finish(); a, b = eng.timing(); k += a; p += b
        res[v].append((k * 100, p * 100))
for v in vals:
    print(f"{name}={v}: kernel " + " ".join(f"{a:.3f}" for a, _ in res[v]) + "   pass " + " ".join(f"{b:.3f}" for _, b in res[v]), "frag", s.n_fragments)
