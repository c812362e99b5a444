#!/usr/bin/env python3
"""Debug aid: run two pileup configurations on one synthetic set and show the first read whose repeats differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
from test_gpu_consistency import SHAPES, PARAMS
si, pi = int(sys.argv[1]), int(sys.argv[2])
va, vb = int(sys.argv[3]), int(sys.argv[4])
o = make_overlaps(device="cuda:0", **SHAPES[si]); p = PARAMS[pi]
cols = (o.read_len,) + o.columns()
res = {}
for v in (va, vb):
    eng = engine.Engine(p, device=0); eng.set_tuning(0, False, v)
    eng.run_device(*cols); s = eng.finish()
    res[v] = {k: x.clone().cpu() for k, x in eng.outputs_device().items()}
    eng.close()
A, B = res[va], res[vb]
print("cov equal", torch.equal(A["cov"], B["cov"]))
ca = A["rep_offset"][1:] - A["rep_offset"][:-1]; cb = B["rep_offset"][1:] - B["rep_offset"][:-1]
bad = torch.nonzero(ca != cb).flatten()
print("reads with different repeat counts:", bad.numel(), bad[:10].tolist())
for r in bad[:3].tolist():
    for name, R in ((va, A), (vb, B)):
        lo, hi = int(R["rep_offset"][r]), int(R["rep_offset"][r + 1])
        print(f"variant {name} read {r} len {int(o.read_len[r])}: repeats", list(zip(R["rep_s"][lo:hi].tolist(), R["rep_e"][lo:hi].tolist())))
    c0, c1 = int(A["cov_offset"][r]), int(A["cov_offset"][r + 1])
    cov = A["cov"][c0:c1]
    hc = int(p.est_cov * p.cov_mul)
    h = (cov >= hc).int().tolist()
    # runs of high windows
    runs = []; st = None
    for i, x in enumerate(h + [0]):
        if x and st is None: st = i
        if not x and st is not None: runs.append((st, i)); st = None
    print(f"  windows {c1-c0} (global {c0}..{c1}), high_cov {hc}, runs of high windows (len>=100):", [(a, b, b - a) for a, b in runs if b - a >= 100][:12])
    Q = 6272
    print("  tile of first window:", c0 // Q, " slot offset in tile ~", c0 % Q)
