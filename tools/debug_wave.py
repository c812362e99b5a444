"""Debug aid (GPU): the wave kernel (variant 5) against the oracle on golden and synthetic cases; prints first differences."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raft_testlib import oracle_run, RaftParams
from test_gpu_parity import load_case
from raft_amd import engine
from raft_amd.synth import make_overlaps

def check(name, p, cols, variant=5, tile=0):
    want = oracle_run(p, *cols)
    eng = engine.Engine(p); eng.set_tuning(tile, False, variant); eng.run_host(*cols); s = eng.finish(); got = eng.fetch()
    off = want["cov_offset"]
    bad = np.flatnonzero(got["cov"] != want["cov"])
    print(f"== {name}: reads {len(cols[0])} rec {len(cols[1])} bins {s.n_bins} sym {s.symmetric} path {s.interval_path} bad windows {bad.size}; "
          f"tot_cov {s.total_coverage}/{want['total_coverage']} tot_rep {s.total_repeat_length}/{want['total_repeat_length']} reps {s.n_repeats}/{len(want['rep_s'])}")
    if bad.size:
        reads = np.unique(np.searchsorted(off, bad, side="right") - 1)
        for r in reads[:8]:
            lo, hi = off[r], off[r + 1]
            b = bad[(bad >= lo) & (bad < hi)]
            print(f"   read {r}: windows [{lo},{hi}) len {cols[0][r]} bad {b.size} first {b[0]-lo} last {b[-1]-lo} got {got['cov'][b[:8]]} want {want['cov'][b[:8]]}")
    for k in ("rep_offset", "rep_s", "rep_e", "cuts", "frag_begin", "frag_end"):
        if not np.array_equal(got[k], want[k]):
            g, w = got[k], want[k]
            n = min(len(g), len(w))
            d = np.flatnonzero(g[:n] != w[:n])
            print(f"   {k} differs: len {len(g)}/{len(w)} first diff at {d[0] if d.size else n}: got {g[d[:5]] if d.size else g[n:n+5]} want {w[d[:5]] if d.size else w[n:n+5]}")
            if k == "rep_offset" and d.size:
                r = d[0] - 1
                print(f"      read {r} len {cols[0][r]} windows {off[r+1]-off[r]} got reps {list(zip(got['rep_s'][g[r]:g[r+1]], got['rep_e'][g[r]:g[r+1]]))} want {list(zip(want['rep_s'][w[r]:w[r+1]], want['rep_e'][w[r]:w[r+1]]))}")
                hc = want["high_cov"]
                cv = want["cov"][off[r]:off[r+1]]
                hi_w = np.flatnonzero(cv >= hc)
                print(f"      high windows of that read: {hi_w[:40]} ... cov_off {off[r]} (mod 512 = {off[r] % 512})")
    eng.close()

which = sys.argv[1:] or ["edge_reads", "g2", "s300_sym", "synth"]
for w in which:
    if w == "synth":
        for n, seed in ((300, 3), (3000, 4), (30000, 5)):
            o = make_overlaps(n, seed=seed)
            cols = [c.numpy() for c in (o.read_len,) + o.columns()]
            check(f"synth{n}", RaftParams(est_cov=30), cols)
    else:
        try:
            p, cols, exp, meta = load_case(w)
        except Exception as e:
            print("no case", w, e); continue
        check(w, p, cols)
