#!/usr/bin/env python3
"""tools/full_compare.py -- the WHOLE result of a full-size pass against the CPU oracle, bit for bit.

BASELINE configs[2] (3.3 M reads, 2.9e8 records, 2e9 windows) and configs[4] (400 k ultralong reads) are too large for one
oracle call inside pytest, so the pytest suites compare samples of them (60 k reads) and invariants.  This tool closes the
gap outside pytest: the engine's full-size result is compared with oracle_run() over EVERY read, in consecutive windows of
reads cut out of the set as closed problems (raft_amd.synth.query_window: in a symmetric PAF a read's outputs depend only on
the records whose query it is, repeat.hpp:48-58) -- coverage, repeats, cut points, fragment bounds / read_num, and the four
stdout sums accumulated over the windows.  Where oracle/_ref/libraft_ref.so is present (built from /root/reference in the
build container; travels to the GPU box) the unmodified reference's own profileCoverage / repeat_annotate run on a sample of
the windows as well.

usage: python tools/full_compare.py [--workload hg002|ultralong|s50k] [--reads N] [--window 150000] [--forms columns,grouped,windows,delta4]
Test infrastructure: uses oracle/ as the checker, never as the thing measured.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    import torch
    import raft_testlib as tl
    from bench import DEFAULT_READS, WORKLOADS
    from raft_amd import engine, hostio
    from raft_amd.params import RaftParams
    from raft_amd.synth import make_overlaps, query_window

    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="hg002", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--window", type=int, default=150_000, help="reads per oracle call")
    ap.add_argument("--forms", default="columns,grouped,windows,delta4")
    ap.add_argument("--ref-windows", type=int, default=2, help="windows that also go through the unmodified reference's code (libraft_ref.so)")
    args = ap.parse_args()

    gen_kw, est_cov, text = WORKLOADS[args.workload]
    n_reads = args.reads or DEFAULT_READS[args.workload]
    p = RaftParams(est_cov=est_cov)
    p_sym = RaftParams(**dict(p.__dict__, symmetric_mode=1))
    dev = "cuda:0"
    o = make_overlaps(n_reads, seed=args.seed, device=dev, **gen_kw)
    torch.cuda.synchronize()
    print(f"full_compare: workload {args.workload}: {o.n_reads} reads, {o.n_rec} records; windows of {args.window} reads", flush=True)

    # ---- the engine's results, one per input form, brought to the host once
    forms = [f for f in args.forms.split(",") if f]
    results = {}
    rl_np = o.read_len.cpu().numpy()
    off = hostio.group_offsets(o.n_reads, o.qid.cpu().numpy(), max_runs=4)
    n_bins = int(((o.read_len.long() + p.reso - 1) // p.reso).sum())
    for form in forms:
        t0 = time.perf_counter()
        e = engine.Engine(p if form == "columns" else p_sym, device=0)
        e.use_torch_stream()
        if form == "columns":
            e.run_device(o.read_len, *o.columns())
        elif form == "grouped":
            e.run_device_grouped(o.read_len, torch.as_tensor(off).to(dev), o.qid, o.qs, o.qe, n_bins=n_bins)
        else:
            win = hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), p.reso)
            if win is None:
                print(f"  form {form}: reads beyond 65,535 windows, window records not possible: skipped")
                continue
            e.set_output_width(8 if form == "delta4" else 1 if est_cov < 40 else 2)
            e.run_device_windows(o.read_len, torch.as_tensor(off).to(dev), torch.as_tensor(win.view(np.int32)).to(dev), n_bins=n_bins)
        s = e.finish()
        got = e.fetch()                         # (decodes the encodings on the device where the pass wrote one)
        got.update(symmetric=s.symmetric, high_cov=s.high_cov, total_coverage=s.total_coverage, total_windows=s.total_windows,
                   total_repeat_length=s.total_repeat_length, total_read_length=s.total_read_length)
        results[form] = got
        e.close()
        print(f"  form {form:8s}: pass + fetch {time.perf_counter() - t0:6.1f} s; {s.n_bins} windows, {s.n_repeats} repeats, {s.n_cuts} cut points, "
              f"{s.n_fragments} fragments, symmetric {s.symmetric}", flush=True)
    if not results:
        raise SystemExit("no form ran")
    first = next(iter(results))
    # the forms against each other, whole arrays
    for f, g in results.items():
        if f == first:
            continue
        for k in tl.ARRAY_KEYS + tl.SCALAR_KEYS:
            assert np.array_equal(np.asarray(g[k]), np.asarray(results[first][k])), f"form {f} differs from {first} in {k}"
        print(f"  form {f} == form {first}: every array and scalar", flush=True)
    got = results[first]
    for f in list(results):
        if f != first:
            del results[f]

    # ---- the oracle over every read, window by window
    tot = {"total_coverage": 0, "total_windows": 0, "total_repeat_length": 0, "total_read_length": 0}
    n_win = (o.n_reads + args.window - 1) // args.window
    t_or = 0.0
    checked = {"cov": 0, "rep": 0, "cuts": 0, "frag": 0}
    ref_done = 0
    for w in range(n_win):
        a, b = w * args.window, min(o.n_reads, (w + 1) * args.window)
        qw = query_window(o, a, b)
        cols = [c.cpu().numpy() for c in (qw.read_len,) + qw.columns()]
        t0 = time.perf_counter()
        want = tl.oracle_run(p, *cols)
        t_or += time.perf_counter() - t0
        n = b - a
        assert want["symmetric"] == 1
        for key, off_key, arrs in (("cov", "cov_offset", ("cov",)), ("rep", "rep_offset", ("rep_s", "rep_e")), ("cuts", "cut_offset", ("cuts",)),
                                   ("frag", "frag_offset", ("frag_begin", "frag_end"))):
            wo = want[off_key][: n + 1]
            go = got[off_key][a: b + 1]
            assert np.array_equal(go - go[0], wo - wo[0]), f"window {w}: {off_key} differs"
            for arr in arrs:
                g = got[arr][go[0]: go[-1]]
                x = want[arr][wo[0]: wo[n]]
                if not np.array_equal(g, x):
                    bad = np.flatnonzero(g != x)
                    raise SystemExit(f"window {w} reads [{a}, {b}): {arr} differs at {bad.size} entries; first at {bad[0]}: got {g[bad[0]]} want {x[bad[0]]}")
            checked[key] += int(go[-1] - go[0])
        # read_num: fragment rows are numbered over the whole set (chop.hpp:195); frag_read names the read
        fo = got["frag_offset"]
        fr = got["frag_read"][fo[a]: fo[b]]
        assert np.array_equal(fr, np.repeat(np.arange(a, b, dtype=np.int32), np.diff(fo[a: b + 1]).astype(np.int64))), f"window {w}: frag_read"
        # the window's share of the stdout sums: the dummy read (index n) holds no query-side records but counts windows / length
        co = want["cov_offset"]
        tot["total_coverage"] += int(want["cov"][: co[n]].astype(np.int64).sum())
        tot["total_windows"] += int(co[n])
        tot["total_read_length"] += int(cols[0][:n].astype(np.int64).sum())
        # unclamped repeat bases are not an output array: taken from a second oracle call's total when the dummy read has none
        dummy_rep = int(want["rep_offset"][n + 1] - want["rep_offset"][n])
        assert dummy_rep == 0, "the dummy read of a window holds a repeat: total_repeat_length cannot be split"
        tot["total_repeat_length"] += want["total_repeat_length"]
        if ref_done < args.ref_windows and tl.have_ref_lib() and (w == 0 or w == n_win // 2):
            r = tl.ref_lib_run(p, *cols, want_cov=True)
            for k in ("cov", "rep_s", "rep_e"):
                assert np.array_equal(np.asarray(r[k]), want[k]), f"window {w}: the reference's own code differs from the oracle in {k}"
            ref_done += 1
            print(f"  window {w}: the unmodified reference's profileCoverage / repeat_annotate agree as well", flush=True)
        print(f"  window {w + 1}/{n_win}: reads [{a}, {b}) ok ({t_or:.0f} s of oracle so far)", flush=True)
        del want, qw, cols
    for k, v in tot.items():
        assert int(got[k]) == v, f"{k}: engine {got[k]} oracle windows {v}"
    print(f"full_compare: OK -- {o.n_reads} reads: {checked['cov']} windows, {checked['rep']} repeats, {checked['cuts']} cut points, "
          f"{checked['frag']} fragments bit-identical to the oracle; stdout sums {tot}; oracle time {t_or:.0f} s")


if __name__ == "__main__":
    main()
