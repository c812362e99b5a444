#!/usr/bin/env python3
"""One chunked end-to-end pass of the bench workload with RAFT_PIPE_TRACE=1 (stage clock per chunk on stderr).
usage: pipe_trace.py [reads] [chunks] [columns|columns_d4|grouped|windows|windows_d4]   (_d4: coverage back as four-bit steps; columns: the engine
derives offsets and window records itself)"""
import os, sys, time
os.environ["RAFT_PIPE_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
from raft_amd import engine, hostio
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3_300_000
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mode = sys.argv[3] if len(sys.argv) > 3 else "columns"
o = make_overlaps(reads, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
host = [c.cpu().pin_memory().numpy() for c in (o.read_len, o.qid, o.qs, o.qe)]
eng = engine.Engine(RaftParams(est_cov=32, symmetric_mode=1))
out = eng.host_output_buffers(host[0], pinned=True, width=8 if mode.endswith("_d4") else 1)
if not mode.startswith("columns"):
    off = hostio.group_offsets(reads, host[1], out=torch.empty(4 * (reads + 1), dtype=torch.int64, pin_memory=True).numpy())
    win = hostio.pack_windows(host[2], host[3], 50, out=torch.empty(o.n_rec, dtype=torch.int32, pin_memory=True).numpy().view(np.uint32))
for it in range(int(os.environ.get("RAFT_TRACE_PASSES", "3"))):
    sys.stderr.write(f"---- pass {it}\n")
    t = time.perf_counter()
    if mode.startswith("columns"):
        res, s = eng.run_pipelined(*host, n_chunks=chunks, out=out)
    elif mode == "grouped":
        res, s = eng.run_pipelined_grouped(host[0], off, host[2], host[3], n_chunks=chunks, out=out)
    else:
        res, s = eng.run_pipelined_windows(host[0], off, win, n_chunks=chunks, out=out)
        if it == 2:
            sys.stderr.write(f"exceptions {res['exc_index'].size} of {s.n_bins} windows\n")
    dt = time.perf_counter() - t
    sys.stderr.write(f"pass {it}: {dt*1e3:.1f} ms -> {o.n_rec/dt:.3e} records/s\n")
