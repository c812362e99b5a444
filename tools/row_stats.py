#!/usr/bin/env python3
"""How do 256-window rows of the benchmark's coverage array look to the run scan?  (sizes the scan's fast paths)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
o = make_overlaps(reads, mean_len=30000.0, coverage=32.0, seed=20242008, device="cuda:0")
eng = engine.Engine(RaftParams(est_cov=32))
eng.run_device(o.read_len, *o.columns()); s = eng.finish()
out = eng.outputs_device()
cov = out["cov"] if isinstance(out, dict) else out.cov
hc = 48
n = (cov.numel() // 256) * 256
h = (cov[:n] >= hc).view(-1, 64, 4)
anyh = h.any(2).any(1)
allh = h.all(2).all(1)
full_lane = h.all(2).any(1)
last = h[:, 63, 3]
print("rows", h.shape[0], "high windows %.4f" % h.float().mean().item())
print("rows with any high window      %.4f" % anyh.float().mean().item())
print("rows all high                  %.4f" % allh.float().mean().item())
print("rows with a fully-high lane    %.4f" % full_lane.float().mean().item())
print("rows any-high, no full lane, last slot low %.4f" % (anyh & ~full_lane & ~last).float().mean().item())
print("mean coverage %.2f, p99 %.0f" % (cov.float().mean().item(), torch.quantile(cov[:5_000_000].float(), 0.99).item()))
