#!/usr/bin/env python3
"""The quantum tile (windows per range a worker draws, and per boundary tile_desc_kernel searches for) swept inside ONE process and
ONE context: kernel and pass per value.  usage: quantum_probe.py q1,q2,... [reps]   env: PROBE_READS=n  (0 = the library's choice)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
vals = [int(v) for v in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
o = make_overlaps(int(os.environ.get("PROBE_READS", "3300000")), mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
eng = engine.Engine(RaftParams(est_cov=32, symmetric_mode=-1))
cols = tuple(eng.device_copy(c) for c in (o.read_len,) + o.columns())
run = lambda: eng.run_device(*cols)
res = {v: [] for v in vals}
sig = {}
for v in vals:
    eng.set_tuning(v, False, -1)
    for _ in range(3):
        run(); s = eng.finish()
    sig[v] = (s.n_fragments, s.total_coverage, s.n_repeats, s.total_repeat_length)
for r in range(reps):
    for v in vals:
        eng.set_tuning(v, False, -1)
        k = pp = 0.0
        for _ in range(10):
            run(); s = eng.finish(); a, b = eng.timing(); k += a; pp += b
        res[v].append((k * 100, pp * 100))
for v in vals:
    print(f"quantum={v}: kernel " + " ".join(f"{a:.3f}" for a, _ in res[v]) + "   pass " + " ".join(f"{b:.3f}" for _, b in res[v]), "sig", sig[v])
assert len(set(sig.values())) == 1, "quanta disagree"
