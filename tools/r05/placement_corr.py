#!/usr/bin/env python3
"""Does a plain store stream into a context's coverage array predict its pileup kernel's time?  N contexts in ONE process (same
inputs), for each: torch fill_ of the cov array (a store stream, events) and the headline pass's kernel time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
policies = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [None]
o = make_overlaps(3300000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
p = RaftParams(est_cov=32)
mem = engine.Engine(p)
cols = tuple(mem.device_copy(c) for c in (o.read_len,) + o.columns())
rows = []
for i in range(n_ctx):
    pol = policies[i % len(policies)]
    if pol is not None:
        engine.set_placement(pol)
    e = engine.Engine(p)
    e.use_torch_stream()
    for _ in range(3):
        e.run_device(*cols); e.finish()
    k = 0.0
    for _ in range(8):
        e.run_device(*cols); e.finish(); a, b = e.timing(); k += a
    cov = e.outputs_device()["cov"]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); ev0.record(); cov.fill_(1); ev1.record(); torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    ev0.record(); s = int(cov[::4096].sum()); ev1.record(); torch.cuda.synchronize()
    rows.append((i, k / 8 * 1e3, best, cov.numel() * 4 / best / 1e6))
    print(f"context {i} policy {pol}: kernel {k / 8 * 1e3:.3f} ms   fill {best:.3f} ms = {cov.numel() * 4 / best / 1e6:.0f} GB/s", flush=True)
    e.close()
