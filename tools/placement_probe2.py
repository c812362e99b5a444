#!/usr/bin/env python3
"""Which of a context's buffers decides whether its pileup kernel runs at 2.2 or at 2.6 ms?  Finds a fast and a slow context in one
process, then swaps their scratch buffers one at a time and times both again.  Needs a library built with
make -C raft_amd/csrc DEFS=-DRAFT_DEBUG_SWAP (raft_hip_debug_swap is not part of the ABI); RAFT_VMM_SPREAD=1 shows the effect best."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raft_amd import engine
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
o = make_overlaps(3_300_000, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
cols = (o.read_len,) + o.columns()
names = "wave_ctr ctrl block_sums tile_cuts cov_off rep_res_off cutcap_off rep_cnt raw_key raw_s raw_e cov tile_first tile_desc scan_tmp samples slow_list cut_cnt frag_cnt rep_off cut_off frag_off rep_s rep_e cuts frag_read frag_begin frag_end".split()
def t(eng):
    for _ in range(2):
        eng.run_device(*cols); eng.finish()
    k = 0.0
    for _ in range(6):
        eng.run_device(*cols); eng.finish(); k += eng.timing()[0]
    return k / 6 * 1e3
engs, times = [], []
for i in range(12):
    e = engine.Engine(RaftParams(est_cov=32)); engs.append(e); times.append(t(e))
    print(f"context {i}: {times[-1]:.3f}", flush=True)
    if max(times) - min(times) > 0.25 and len(times) >= 2:
        break
fast, slow = engs[times.index(min(times))], engs[times.index(max(times))]
lib = fast._lib
lib.raft_hip_debug_swap.argtypes = [engine.C.c_void_p, engine.C.c_void_p, engine.C.c_int]
print(f"fast {t(fast):.3f} slow {t(slow):.3f}")
for w, nm in enumerate(names):
    lib.raft_hip_debug_swap(fast._ctx, slow._ctx, w)
    a, b = t(fast), t(slow)
    print(f"swapped {nm:12s}: 'fast' {a:.3f} 'slow' {b:.3f}" + ("   <-- moved" if a > b + 0.15 else ""), flush=True)
    lib.raft_hip_debug_swap(fast._ctx, slow._ctx, w)
os._exit(0)
