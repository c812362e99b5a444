#!/usr/bin/env python3
"""Where does the process-to-process spread of the pileup kernel's time come from?  One process: several contexts (the engine's
own buffers land elsewhere each time) over the same inputs, then several copies of the inputs under one context."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
o = make_overlaps(3_300_000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
cols = (o.read_len,) + o.columns()
def t(eng, c):
    for _ in range(3):
        eng.run_device(*c); eng.finish()
    k = 0.0
    for _ in range(10):
        eng.run_device(*c); eng.finish(); k += eng.timing()[0]
    return k * 100
engs = []
for i in range(int(os.environ.get("N_CTX", "10"))):
    e = engine.Engine(RaftParams(est_cov=32)); engs.append(e)
    print(f"context {i}: kernel {t(e, cols):.3f} ms", flush=True)

print(f"context 0 again: {t(engs[0], cols):.3f}   context 3 again: {t(engs[3], cols):.3f}")
keep = []
for i in range(6):
    c2 = tuple(x.clone() for x in cols); keep.append(c2)
    print(f"input copy {i} (torch): context 0 kernel {t(engs[0], c2):.3f}   context 3 kernel {t(engs[3], c2):.3f}", flush=True)
for i in range(3):
    c3 = tuple(engs[0].device_copy(x) for x in cols); keep.append(c3)
    print(f"input copy {i} (raft_hip_device_alloc): context 0 kernel {t(engs[0], c3):.3f}   context 3 kernel {t(engs[3], c3):.3f}", flush=True)
