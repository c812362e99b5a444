#!/usr/bin/env python3
"""PCIe-inclusive rate of one pass: int32 columns in host memory (pageable, then page-locked and
reused) -> engine -> every output back on the host.  This is NOT bench.py's `value` (which starts with the inputs resident in HBM)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3_300_000
o = make_overlaps(reads, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
eng = engine.Engine(RaftParams(est_cov=32))
for kind in ("pageable", "pinned"):
    if kind == "pageable":
        host = [c.cpu().numpy() for c in (o.read_len,) + o.columns()]
    else:
        host = [c.cpu().pin_memory().numpy() for c in (o.read_len,) + o.columns()]
    out = None
    for it in range(3):
        t0 = time.perf_counter()
        eng.run_host(*host)
        s = eng.finish()
        t1 = time.perf_counter()
        out = eng.fetch(pinned=(kind == "pinned"), out=out if kind == "pinned" else None)
        t2 = time.perf_counter()
        in_gb = sum(a.nbytes for a in host) / 1e9
        out_gb = sum(a.nbytes for a in out.values()) / 1e9
        print(f"{kind} iter {it}: H2D+pass {t1-t0:.3f} s ({in_gb:.2f} GB in), D2H {t2-t1:.3f} s ({out_gb:.2f} GB out), "
              f"end-to-end {o.n_rec/(t2-t0):.3e} PAF records/s, {s.n_fragments/(t2-t0):.3e} fragments/s  [{kind} host memory]")
