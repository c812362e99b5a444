#!/usr/bin/env python3
"""Differential-fuzz fixture: random small TEXT inputs run through the compiled reference binary.

Build container only (needs oracle/_ref/raft, built by `make -C oracle` from /root/reference).  Every case is
written as FASTA + PAF text, run through the unmodified `raft` binary, and the reference's three result files are
parsed back into integer arrays.  Unlike the in-process harness (oracle/ref_harness.cpp), the binary also runs
break_reads(), so this pins fragment bounds and read_num (chop.hpp:193-324) to the reference on hundreds of inputs,
not only on the hand-written golden cases.

The fixture is data only -- inputs (int32 columns) and the reference's outputs for them -- stored as one compressed
.npz (ref_fuzz.npz) with CSR offsets per case.  What the generator aims at:
  * -l not a multiple of -p, -l == -p, -l a large multiple (div from 1 to 5)
  * -v from 0 up to F[div]'s lower bound div*L (the largest value for which the reference is defined on every input)
  * empty and inverted intervals, self overlaps, intervals ending inside the last partial window
  * the symmetric flag flipping mid-stream (mirror of record 0 planted at a random position) or staying 0
  * reads without overlaps, 0- and 1-base reads, reso in {1, 7, 50, 64}
  * repeats that clamp to start 0 in reads with more than 16 repeats (the std::sort tie corner, repeat.hpp:170)
  * dense high-coverage stretches so that long_repeats and masked markers are non-trivial

Usage:  python tests/golden/make_ref_fuzz.py [n_cases]
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from raft_testlib import REF_BIN, RaftParams, md5, result_from_ref_files, run_ref_binary, tie_case, write_fasta, write_paf  # noqa: E402

OUT_FILES = ("reads.fasta", "coverage.txt", "long_repeats.txt", "long_repeats.bed")


def fuzz_case(seed: int):
    """-> (RaftParams, [read_len, qid, qs, qe, tid, ts, te]).  Deterministic in `seed`."""
    if seed % 8 == 7:                                   # the sort-tie shape, through the binary this time
        p, cols = tie_case(seed)
        p.interval_length = p.repeat_length              # the CLI sets both with -p (main.cpp:44-47)
        p.read_length = p.repeat_length * 3 + 1
        p.overlap_length = int(np.random.default_rng(seed).integers(0, p.interval_length + 1))
        return p, cols
    rng = np.random.default_rng(50_000 + seed)
    n = int(rng.integers(1, 40))
    reso = int(rng.choice([1, 7, 50, 64]))
    L = int(rng.choice([60, 100, 250, 1000]))
    len_hi = int(rng.choice([400, 3000, 12000]))
    rl = rng.integers(0, len_hi, n).astype(np.int32)
    if seed % 5 == 0:
        rl[rng.integers(0, n)] = rng.integers(0, 2)      # 0- and 1-base reads
    m = int(rng.integers(1, 400))
    qid = rng.integers(0, n, m).astype(np.int32)
    tid = rng.integers(0, n, m).astype(np.int32)
    if seed % 3 == 0:
        qid.sort()
    if seed % 4 == 1:                                   # pile most records onto a few reads: long high-coverage runs
        hot = rng.integers(0, n, 3)
        sel = rng.random(m) < 0.7
        qid[sel] = rng.choice(hot, int(sel.sum()))

    def coords(ids, wide):
        ln = rl[ids].astype(np.int64)
        hi = ((ln + reso - 1) // reso) * reso           # any end <= nb*reso stays inside the last window (defined)
        a = (rng.random(m) * (hi + 1)).astype(np.int64)
        b = (rng.random(m) * (hi + 1)).astype(np.int64)
        s, e = np.minimum(a, b), np.maximum(a, b)
        if wide:                                        # long intervals: whole stretches above high_cov
            s = (s * 0.3).astype(np.int64)
            e = np.minimum(hi, e + (0.5 * ln).astype(np.int64))
        inv = rng.random(m) < 0.08                      # some inverted / empty intervals
        return np.where(inv, e, s).astype(np.int32), np.where(inv, s, e).astype(np.int32)

    qs, qe = coords(qid, seed % 2 == 1)
    ts, te = coords(tid, False)
    if seed % 2 == 0 and m > 3:                          # plant the mirror of record 0 -> the flag flips mid-stream
        k = int(rng.integers(1, m))
        qid[k], tid[k], qs[k], qe[k], ts[k], te[k] = tid[0], qid[0], ts[0], te[0], qs[0], qe[0]
    div = int(rng.integers(1, 6))
    read_length = L * div + (int(rng.integers(0, L)) if seed % 3 else 0)   # -l a multiple of -p in a third of the cases
    v_mode = seed % 4
    overlap = (0, int(rng.integers(0, L)), div * L, int(rng.integers(0, div * L + 1)))[v_mode]
    p = RaftParams(reso=reso, est_cov=int(rng.integers(1, 6)), cov_mul=float(rng.choice([1.0, 1.3, 1.5, 2.0])),
                   repeat_length=L, interval_length=L, read_length=read_length, overlap_length=overlap,
                   flanking_length=int(rng.choice([0, 30, 300, 5000])))
    return p, [rl, qid, qs, qe, tid, ts, te]


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    assert os.path.exists(REF_BIN), "oracle/_ref/raft missing: run `make -C oracle` in the build container"
    P, scal, md = [], [], []
    cat = {k: [] for k in ("read_len", "qid", "qs", "qe", "tid", "ts", "te", "cov", "rep_cnt", "rep_s", "rep_e",
                           "frag_read", "frag_begin", "frag_end")}
    offs = {k: [0] for k in ("reads", "recs", "cov", "rep", "frag")}
    kept = []
    for seed in range(n_cases):
        p, cols = fuzz_case(seed)
        names = [f"r{i}" for i in range(len(cols[0]))]
        with tempfile.TemporaryDirectory() as tmp:
            write_fasta(os.path.join(tmp, "reads.fa"), names, cols[0])
            write_paf(os.path.join(tmp, "overlaps.paf"), names, *cols)
            rc, out = run_ref_binary(tmp, p.cli_args() + ["-o", "out"], "reads.fa", "overlaps.paf")
            assert rc == 0, (seed, rc, out[-300:])
            res = result_from_ref_files(os.path.join(tmp, "out"), names)
            digests = [md5(open(os.path.join(tmp, "out." + f), "rb").read()) for f in OUT_FILES]
        text = out.decode()
        sym = int("INFO, Symmetric overlaps 1 " in text)
        stats = [l for l in text.split("\n") if l.startswith(("coverage per window", "fraction_of_repeat_length"))]
        kept.append(seed)
        P.append([p.reso, p.est_cov, p.repeat_length, p.interval_length, p.read_length, p.overlap_length, p.flanking_length])
        scal.append((p.cov_mul, sym, "\n".join(stats)))
        md.append(digests)
        for k, a in zip(("read_len", "qid", "qs", "qe", "tid", "ts", "te"), cols):
            cat[k].append(np.asarray(a, np.int32))
        cat["cov"].append(res["cov"])
        cat["rep_cnt"].append(np.diff(res["rep_offset"]).astype(np.int32))
        for k in ("rep_s", "rep_e", "frag_read", "frag_begin", "frag_end"):
            cat[k].append(res[k])
        offs["reads"].append(offs["reads"][-1] + len(cols[0]))
        offs["recs"].append(offs["recs"][-1] + len(cols[1]))
        offs["cov"].append(offs["cov"][-1] + len(res["cov"]))
        offs["rep"].append(offs["rep"][-1] + len(res["rep_s"]))
        offs["frag"].append(offs["frag"][-1] + len(res["frag_read"]))
    arrays = {k: (np.concatenate(v) if v else np.empty(0, np.int32)).astype(np.int32) for k, v in cat.items()}
    np.savez_compressed(os.path.join(HERE, "ref_fuzz.npz"), seeds=np.array(kept, np.int32), params=np.array(P, np.int32),
                        cov_mul=np.array([s[0] for s in scal], np.float64), symmetric=np.array([s[1] for s in scal], np.int32),
                        stats=np.array([s[2] for s in scal]), md5=np.array(md),
                        **{"off_" + k: np.array(v, np.int64) for k, v in offs.items()}, **arrays)
    n_multi = sum(1 for i in range(len(kept)) if offs["frag"][i + 1] - offs["frag"][i] > offs["reads"][i + 1] - offs["reads"][i])
    print(f"ref_fuzz.npz: {len(kept)} cases, {offs['recs'][-1]} records, {offs['cov'][-1]} windows, {offs['rep'][-1]} repeats, "
          f"{offs['frag'][-1]} fragments ({n_multi} cases split at least one read), "
          f"{int(np.sum(arrays['rep_cnt'] > 16))} reads with > 16 repeats, "
          f"{os.path.getsize(os.path.join(HERE, 'ref_fuzz.npz')) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
