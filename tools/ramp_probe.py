#!/usr/bin/env python3
"""Does the pass get faster while the process runs (clocks, page tables)?  Times the bench set's six-column pass in groups of ten."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
t00 = time.perf_counter()
o = make_overlaps(3_300_000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
torch.cuda.synchronize()
print(f"workload ready at {time.perf_counter() - t00:.2f} s")
eng = engine.Engine(RaftParams(est_cov=32))
cols = (o.read_len,) + o.columns()
for g in range(int(os.environ.get("RAMP_GROUPS", "30"))):
    torch.cuda.synchronize(); t = time.perf_counter(); k = 0.0
    for _ in range(10):
        eng.run_device(*cols); eng.finish(); k += eng.timing()[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"t={time.perf_counter() - t00:6.2f} s  passes {g*10:3d}..{g*10+9:3d}: {dt*100:.3f} ms/pass  kernel {k*100:.3f} ms")
    if g == 14:
        time.sleep(2.0); print("(slept 2 s)")
