#!/usr/bin/env python3
"""What a context's first pass pays for its buffers (DevBuf: chunks spread over physical memory): first and second pass, wall clock."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
o = make_overlaps(3_300_000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
cols = (o.read_len,) + o.columns()
torch.cuda.synchronize()
for i in range(3):
    e = engine.Engine(RaftParams(est_cov=32))
    t = time.perf_counter(); e.run_device(*cols); e.finish(); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); e.run_device(*cols); e.finish(); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    print(f"context {i}: first pass {t1*1e3:.1f} ms, second {t2*1e3:.2f} ms (kernel {e.timing()[0]*1e3:.3f})", flush=True)
    e.close()
