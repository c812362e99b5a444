import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raft_testlib import oracle_run
from test_gpu_parity import random_case
from raft_amd import engine
seed, tile, variant = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
p, cols = random_case(seed)
want = oracle_run(p, *cols)
eng = engine.Engine(p); eng.set_tuning(tile, False, variant); eng.run_host(*cols); s = eng.finish(); got = eng.fetch()
print("params", p, "n_reads", len(cols[0]), "n_rec", len(cols[1]), "bins", s.n_bins, "sym", s.symmetric, "path", s.interval_path)
off = want["cov_offset"]
bad = np.flatnonzero(got["cov"] != want["cov"])
print("bad windows", bad.size)
if bad.size:
    reads = np.unique(np.searchsorted(off, bad, side="right") - 1)
    for r in reads[:10]:
        lo, hi = off[r], off[r + 1]
        b = bad[(bad >= lo) & (bad < hi)]
        print(f" read {r}: windows [{lo},{hi}) len {cols[0][r]} bad {b.size} first {b[0]-lo} last {b[-1]-lo} tile_of_start {lo//max(tile,1)} got {got['cov'][b[:6]]} want {want['cov'][b[:6]]}")
for k in ("rep_s", "rep_e", "cuts", "frag_begin"):
    print(k, np.array_equal(got[k], want[k]))
