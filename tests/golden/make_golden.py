#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by RUNNING THE COMPILED REFERENCE.

Only works in the build container (needs oracle/_ref/raft, built by `make -C oracle`
from /root/reference).  The fixtures are data: inputs (int32 columns / literal FASTA+PAF
text written here) and the reference's outputs for them; no reference source is stored.

  micro/<case>/      literal text inputs (the survey's hand-traced vectors G1-G3 and reader
                     edge cases G4) + the reference's four output files + its stdout
  <case>.npz         seeded synthetic SoA inputs + outputs parsed from the reference's files,
                     plus md5 digests of the reference's output files for the CLI test
  manifest.json      parameters and digests per case

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import gzip
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from raft_testlib import (REF_BIN, RaftParams, md5, result_from_ref_files, run_ref_binary, seq_of, write_fasta,  # noqa: E402
                          write_paf)

from raft_amd.synth import make_overlaps  # noqa: E402

OUT_FILES = ("reads.fasta", "coverage.txt", "long_repeats.txt", "long_repeats.bed")


def strip_timing(stdout: bytes) -> str:
    keep = []
    for line in stdout.decode().split("\n"):
        if line.startswith("INFO, main(), program completed after") or line.startswith("INFO, main(), CMD:"):
            continue
        keep.append(line)
    return "\n".join(keep)


# ---- literal micro cases ----------------------------------------------------------------------------

def paf_line(q, ql, qs, qe, strand, t, tl, ts, te, extra="10\t20\t60"):
    return f"{q}\t{ql}\t{qs}\t{qe}\t{strand}\t{t}\t{tl}\t{ts}\t{te}\t{extra}\n"


def micro_cases():
    cases = {}
    # G1: binning quirks (empty interval, self overlap, target-side use, short line)
    fa = "".join(f">{n}\n{seq_of(l, i)}\n" for i, (n, l) in enumerate([("rA", 230), ("rB", 100), ("rC", 51)]))
    paf = (paf_line("rA", 230, 0, 120, "+", "rB", 100, 0, 100) + paf_line("rA", 230, 75, 75, "+", "rC", 51, 10, 51) +
           paf_line("rB", 100, 50, 100, "-", "rC", 51, 0, 50) + paf_line("rA", 230, 100, 230, "+", "rA", 230, 0, 130) +
           "short line\n")
    cases["g1"] = (fa, paf, ["-e", "2", "-m", "1.0", "-r", "50", "-p", "100", "-l", "200", "-f", "10"])
    # G2: masked markers, -v back-overlap, 5 + 5 + 1 fragments
    fa = "".join(f">{n}\n{seq_of(l, i)}\n" for i, (n, l) in enumerate([("r0", 1234), ("r1", 900), ("r2", 400)]))
    paf = paf_line("r0", 1234, 0, 1234, "+", "r1", 900, 0, 900) + paf_line("r0", 1234, 300, 560, "+", "r2", 400, 0, 260) * 3
    cases["g2"] = (fa, paf, ["-e", "2", "-m", "1.5", "-r", "50", "-p", "100", "-l", "200", "-f", "30", "-v", "20", "-o", "raft"])
    # G3: symmetric flip + simulated-read headers
    n0 = "read=1,forward,position=1000-2234,length=1234,h1tg000001l"
    n1 = "read=2,reverse,position=5000-5900,length=900,h2tg000002l"
    fa = f">{n0}\n{seq_of(1234)}\n>{n1}\n{seq_of(900, 1)}\n"
    paf = (paf_line(n0, 1234, 300, 900, "+", n1, 900, 0, 600) + paf_line(n0, 1234, 300, 560, "+", n1, 900, 100, 360) * 3 +
           paf_line(n1, 900, 0, 600, "+", n0, 1234, 300, 900) + paf_line(n1, 900, 100, 360, "+", n0, 1234, 300, 560) * 3)
    cases["g3"] = (fa, paf, ["-e", "2", "-m", "1.5", "-r", "50", "-p", "100", "-l", "200", "-f", "30", "-v", "20", "-o", "raft"])
    # G4: reader equivalences -- multi-line FASTA with CRLF/blank lines, header comments, FASTQ-free; PAF with
    # CRLF, >12 columns, exactly 10 columns, 7-field and empty lines
    s0, s1, s2 = seq_of(230), seq_of(100, 1), seq_of(51, 2)
    fa = f">rA desc here\r\n{s0[:100]}\r\n{s0[100:]}\r\n\r\n>rB\tx\n{s1}\n>rC\n{s2[:20]}\n{s2[20:]}\n"
    paf = ("rA\t230\t0\t120\t+\trB\t100\t0\t100\t50\t100\t60\ttp:A:P\tcm:i:5\r\n" + "rA\t230\t75\t75\t+\trC\t51\t10\t51\t1\n" +
           "\n" + "a\tb\tc\td\te\tf\tg\n" + "rB\t100\t50\t100\t-\trC\t51\t0\t50\t1\t1\t60\r\n" +
           "rA\t230\t100\t230\t+\trA\t230\t0\t130\t1\t1\t60")
    cases["g4"] = (fa, paf, ["-e", "1", "-m", "1.3", "-r", "50", "-p", "100", "-l", "200", "-f", "10"])
    # G5: the -v fallthrough renames the outputs (main.cpp:51-55): prefix becomes "20"
    cases["g5_vprefix"] = (cases["g2"][0], cases["g2"][1], ["-e", "2", "-r", "50", "-p", "100", "-l", "200", "-f", "30", "-v", "20"])
    return cases


def make_micro(root):
    man = {}
    for name, (fa, paf, args) in micro_cases().items():
        d = os.path.join(root, "micro", name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        open(os.path.join(d, "reads.fa"), "w", newline="").write(fa)
        open(os.path.join(d, "overlaps.paf"), "w", newline="").write(paf)
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copy(os.path.join(d, "reads.fa"), tmp)
            shutil.copy(os.path.join(d, "overlaps.paf"), tmp)
            rc, out = run_ref_binary(tmp, args, "reads.fa", "overlaps.paf")
            assert rc == 0, (name, rc, out)
            produced = sorted(f for f in os.listdir(tmp) if f not in ("reads.fa", "overlaps.paf"))
            for f in produced:
                shutil.copy(os.path.join(tmp, f), os.path.join(d, "expect." + f))
            open(os.path.join(d, "expect.stdout"), "w").write(strip_timing(out))
        man[name] = {"args": args, "outputs": produced}
        # gz variants of the same inputs must give the same outputs (G4)
    return man


# ---- seeded synthetic cases ---------------------------------------------------------------------------

def synth_cases():
    c = {}
    c["s300_default"] = (dict(n_reads=300, seed=11), RaftParams(est_cov=30))
    c["s300_nonsym_shuffled"] = (dict(n_reads=300, seed=12, symmetric=False, shuffle=True), RaftParams(est_cov=15))
    c["s300_sym_shuffled"] = (dict(n_reads=300, seed=13, shuffle=True), RaftParams(est_cov=30))
    c["s200_smallparams"] = (dict(n_reads=200, seed=14, mean_len=8000, n_families=6, rep_len=(2000, 6000)),
                             RaftParams(reso=7, est_cov=30, cov_mul=1.2, repeat_length=300, interval_length=300,
                                        read_length=1000, overlap_length=20, flanking_length=50))
    c["s60_ultralong"] = (dict(n_reads=60, seed=15, mean_len=150000, sigma=0.8, max_len=900000, coverage=40, n_families=4,
                               copies=4, rep_len=(30000, 60000)),
                          RaftParams(reso=10, est_cov=40, cov_mul=1.5, repeat_length=5000, interval_length=5000,
                                     read_length=20000, overlap_length=500, flanking_length=1000))
    c["s150_reso1"] = (dict(n_reads=150, seed=16, mean_len=3000, min_len=60, max_len=9000, min_ovl=50, n_families=3,
                            rep_len=(400, 900)),
                       RaftParams(reso=1, est_cov=25, cov_mul=1.5, repeat_length=120, interval_length=120,
                                  read_length=500, overlap_length=10, flanking_length=30))
    return c


def edge_case():
    """Hand-built columns: empty / tiny reads, reads without overlaps, empty & inverted intervals,
    self overlaps, intervals ending exactly at len and inside the last partial window."""
    read_len = np.array([0, 1, 49, 50, 51, 230, 1234, 900, 400, 0, 5000, 12000, 7], np.int32)
    rec = [  # qid qs qe tid ts te
        (5, 0, 120, 7, 0, 100), (5, 75, 75, 8, 10, 51), (7, 50, 100, 8, 0, 50), (5, 100, 230, 5, 0, 130),
        (6, 0, 1234, 7, 0, 900), (6, 300, 560, 8, 0, 260), (6, 300, 560, 8, 0, 260), (6, 300, 560, 8, 0, 260),
        (2, 0, 49, 3, 0, 50), (3, 49, 50, 4, 50, 51), (4, 0, 51, 2, 10, 10), (1, 0, 1, 12, 0, 7),
        (10, 100, 4999, 11, 0, 4899), (10, 0, 5000, 11, 7000, 12000), (11, 3000, 9000, 10, 0, 5000),
        (11, 2999, 9001, 10, 60, 40), (11, 0, 12000, 11, 0, 12000), (12, 3, 3, 1, 0, 0), (8, 399, 400, 6, 1233, 1234),
        (10, 2500, 2500, 11, 5000, 5000), (10, 2450, 2450, 11, 5001, 5001),
    ] + [(11, 2000 + 10 * i, 8000 - 10 * i, 10, 500, 4500) for i in range(12)]
    cols = [np.array([r[k] for r in rec], np.int32) for k in range(6)]
    p = RaftParams(reso=50, est_cov=4, cov_mul=1.5, repeat_length=1000, interval_length=1000, read_length=3000,
                   overlap_length=100, flanking_length=200)
    return read_len, cols, p


def run_case(name, read_len, cols, p: RaftParams, root):
    names = [f"r{i}" for i in range(len(read_len))]
    with tempfile.TemporaryDirectory() as tmp:
        write_fasta(os.path.join(tmp, "reads.fa"), names, read_len)
        write_paf(os.path.join(tmp, "overlaps.paf"), names, read_len, *cols)
        args = p.cli_args() + ["-o", "out"]
        rc, out = run_ref_binary(tmp, args, "reads.fa", "overlaps.paf")
        assert rc == 0, (name, rc, out[-400:])
        res = result_from_ref_files(os.path.join(tmp, "out"), names)
        digests = {f: md5(open(os.path.join(tmp, "out." + f), "rb").read()) for f in OUT_FILES}
        stdout = strip_timing(out)
    np.savez_compressed(os.path.join(root, name + ".npz"), read_len=np.asarray(read_len, np.int32),
                        qid=cols[0], qs=cols[1], qe=cols[2], tid=cols[3], ts=cols[4], te=cols[5],
                        **{"exp_" + k: v for k, v in res.items()})
    sym = int("INFO, Symmetric overlaps 1 " in stdout)
    return {"params": p.__dict__, "args": args, "md5": digests, "stdout": stdout, "symmetric": sym,
            "n_reads": int(len(read_len)), "n_rec": int(len(cols[0]))}


def config1_case(root):
    """BASELINE configs[0] (README.md:12-33: `raft -e 42 -o fragmented chr11-2M.fa.gz overlaps.paf`) restated as
    SURVEY.md §8(d) prescribes, because the real chr11 reads are not obtainable offline: G = 2 Mbp, 42x,
    HiFi-like lengths ~N(20 kb, 3 kb), symmetric PAF of all true overlaps >= 500 bp, `-e 42`, and BOTH inputs
    gzip-compressed (the reference reads them through zlib: chop.hpp:93, paf.hpp:29)."""
    gen = dict(n_reads=4200, seed=42, mean_len=20000.0, sigma=0.15, coverage=42.0, min_len=5000, max_len=40000, n_families=3)
    o = make_overlaps(**gen)
    read_len, cols = o.read_len.numpy(), [c.numpy() for c in o.columns()]
    names = [f"m64011_{i}/ccs" for i in range(len(read_len))]
    p = RaftParams(est_cov=42)
    args = ["-e", "42", "-o", "fragmented"]
    with tempfile.TemporaryDirectory() as tmp:
        write_fasta(os.path.join(tmp, "reads.fa"), names, read_len)
        write_paf(os.path.join(tmp, "overlaps.paf"), names, read_len, *cols)
        for f in ("reads.fa", "overlaps.paf"):
            with open(os.path.join(tmp, f), "rb") as i, gzip.open(os.path.join(tmp, f + ".gz"), "wb", compresslevel=1) as z:
                shutil.copyfileobj(i, z)
        rc, out = run_ref_binary(tmp, args, "reads.fa.gz", "overlaps.paf.gz")
        assert rc == 0, (rc, out[-400:])
        res = result_from_ref_files(os.path.join(tmp, "fragmented"), names)
        digests = {f: md5(open(os.path.join(tmp, "fragmented." + f), "rb").read()) for f in OUT_FILES}
        rc2, out2 = run_ref_binary(tmp, ["-e", "42", "-o", "plain"], "reads.fa", "overlaps.paf")   # gz == plain
        assert rc2 == 0 and all(md5(open(os.path.join(tmp, "plain." + f), "rb").read()) == digests[f] for f in OUT_FILES)
        stdout = strip_timing(out)
    np.savez_compressed(os.path.join(root, "c1_chr11_standin.npz"), read_len=read_len.astype(np.int32),
                        qid=cols[0], qs=cols[1], qe=cols[2], tid=cols[3], ts=cols[4], te=cols[5],
                        **{"exp_" + k: v for k, v in res.items()})
    return {"params": p.__dict__, "args": args, "md5": digests, "stdout": stdout, "generator": gen,
            "symmetric": int("INFO, Symmetric overlaps 1 " in stdout), "n_reads": int(len(read_len)),
            "n_rec": int(len(cols[0])), "name_format": "m64011_{i}/ccs", "inputs": ["reads.fa.gz", "overlaps.paf.gz"]}


def main():
    assert os.path.exists(REF_BIN), "oracle/_ref/raft missing: run `make -C oracle` in the build container"
    man = {"micro": make_micro(HERE), "synthetic": {}}
    for name, (gen, p) in synth_cases().items():
        o = make_overlaps(**gen)
        man["synthetic"][name] = run_case(name, o.read_len.numpy(), [c.numpy() for c in o.columns()], p, HERE)
        man["synthetic"][name]["generator"] = gen
    rl, cols, p = edge_case()
    man["synthetic"]["edge_reads"] = run_case("edge_reads", rl, cols, p, HERE)
    man["config1"] = {"c1_chr11_standin": config1_case(HERE)}
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(HERE) for f in fs)
    print(f"golden fixtures written, {total / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
