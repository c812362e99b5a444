#!/usr/bin/env python3
"""Two builds of libraft_hip.so in ONE process, on the same input columns: the pileup kernel's time by HIP events, several contexts
per build (where a context's coverage array lies moves its time by more than most code changes do: DESIGN.md I.4), passes in turn.
usage: lib_ab.py <libA.so> <libB.so> [reads=3300000] [contexts=3] [form=columns|windows] [width=4]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raft_amd import engine, hostio
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps

libs = [os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])]
reads = int(sys.argv[3]) if len(sys.argv) > 3 else 3_300_000
n_ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 3
form = sys.argv[5] if len(sys.argv) > 5 else "columns"
width = int(sys.argv[6]) if len(sys.argv) > 6 else 4
o = make_overlaps(reads, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
p = RaftParams(est_cov=32, symmetric_mode=1 if form == "windows" else -1)
cols = (o.read_len,) + tuple(o.columns())
if form == "windows":
    off = torch.as_tensor(hostio.group_offsets(o.n_reads, o.qid.cpu().numpy(), max_runs=4)).to("cuda:0")
    win = torch.as_tensor(hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), p.reso).view("int32")).to("cuda:0")
    n_bins = int(((o.read_len.long() + p.reso - 1) // p.reso).sum())
engs = []
for li, path in enumerate(libs):
    engine._lib = engine.load_library(path)
    for k in range(n_ctx):
        e = engine.Engine(p, device=0)
        e.set_output_width(width)
        engs.append((li, k, e))
sig = None
times = {(li, k): [] for li, k, _ in engs}
for it in range(8):
    for li, k, e in engs:
        if form == "windows":
            e.run_device_windows(o.read_len, off, win, n_bins=n_bins)
        else:
            e.run_device(*cols)
        s = e.finish()
        t = (s.n_fragments, s.n_repeats, s.total_coverage)
        sig = sig or t
        assert t == sig, (li, k, t, sig)
        if it >= 3:
            times[(li, k)].append(e.timing()[0] * 1e3)
for li, path in enumerate(libs):
    per = [sum(times[(li, k)]) / len(times[(li, k)]) for k in range(n_ctx)]
    print(f"{os.path.basename(path):28s} kernel ms per context: " + " ".join(f"{x:.3f}" for x in per) + f"   min {min(per):.3f} mean {sum(per)/len(per):.3f}")
