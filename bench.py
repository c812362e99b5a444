#!/usr/bin/env python3
"""bench.py -- RAFT hot path (PAF overlaps -> coverage -> repeat mask -> fragments) on MI355X.

One "step" = one full pass of the engine over one synthetic all-vs-all overlap set that is already resident in HBM when the
clock starts: stream inspection (sorted runs, symmetric detection), tile cuts, the pileup / prefix-sum / run-scan kernel,
repeat ordering, cut points (chop.hpp's final_stars), fragment table and the stdout statistics.  Outputs stay in HBM.

Input form (`--input`): `columns` (default, the headline `value`) -- the six plain int32 columns of SURVEY.md §8(b)'s C-ABI
into a detecting context (symmetric_mode = -1): everything create_pileup has to find out about the stream (sorted runs, the
mirror of record 0, which records belong to which read) is found out INSIDE the timed step.  `grouped` / `windows` -- what a
tokeniser that resolves every name can hand over beside the columns (per sorted run, where every read's records begin; the
window count; one word per record): those passes are reported as extra legs (`grouped`, `window_records`) and are never the
headline, because part of create_pileup's bucketing then happens outside the step.

Workload (config.workload), default `hg002`: BASELINE.json configs[2] restated synthetically:
HG002-like 32x set, 3.3 M reads of 30 kb mean length, ~2.9e8 symmetric PAF records written as a cis file followed by a
trans file, each grouped by ascending query id.  Other workloads (never the headline line): `ultralong` = configs[4]
(60x, 150 kb mean, reads up to 1.5 Mb, 50 kb tandem arrays), `s50k` = configs[1] (50 k reads, 20 kb, 30x).

Several GPUs (`--gpus N`, one process per GPU): the headline is BASELINE configs[3] -- the ONE set of configs[2], reads owned
in N contiguous ranges, "scaling": "strong".  By default in its PRE-SPLIT form: every rank holds the rank-th contiguous slice of
the record stream (as window records, the tokeniser's one word per record) and every step routes the records to the owners
of their reads with ONE all-to-all-v over xGMI (raft_hip_exchange: RCCL all-gather of the piece sizes + grouped send /
receive) before the rank's pass; `value` = records of the whole set / step time (max over ranks); the ranks' totals are
checked against a single-GPU pass over the same set.  Beside it, as legs: `host_routed` -- the same set, every rank handed
the records of its reads, no data-path collective (what a tokeniser that knows the owners does; `--host-routed` makes it the
headline) -- and `weak` -- every rank an independent set of the workload's size (rounds 1-5's headline; `--weak`).
Ranks that share a GPU (a one-GPU box, gloo) run the same steps with the exchange in torch: a check of the code path only.
A rank that stalls (a hung collective) is ended by a watchdog (`--watchdog-seconds`): the job exits non-zero instead of hanging.

The line also carries (N = 1): `packed_output` -- the pass exactly as the CLI and the host pipelines run it (grouped, no
query column, coverage written as one byte per window by the pileup kernel itself); `e2e` -- the same workload from
page-locked host columns to every output back on the host (SURVEY.md §8d `t_e2e`; never the headline `value`);
`roofline.pass_frac` -- algorithmic bytes over the device time of the WHOLE pass (cut points included: the pass writes
chop.hpp's final_stars itself; `pass_device_ms_without_cuts` is the same pass with them left to the first fetch);
`cpu_baseline`.  `e2e.records_per_s` starts at SURVEY.md §8(d)'s boundary -- page-locked int32 columns in, every output back
on the host -- with everything the engine's host side derives from the columns inside the clock.

`python bench.py --gpus N` without a launcher starts the N ranks itself (child processes, before any GPU call); under
`python -m torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (generator keywords, est_cov, description)
    "hg002": (dict(mean_len=30000.0, coverage=32.0), 32,
              "HG002-like 32x all-vs-all PAF restated synthetically (BASELINE configs[2]): cis+trans files grouped by query "
              "id, symmetric, raft -e 32 defaults (-r 50 -p 10000 -l 20000 -f 1000 -v 500)"),
    "ultralong": (dict(mean_len=150000.0, coverage=60.0, sigma=0.7, min_len=10000, max_len=1_500_000, copies=6,
                       rep_len=(45000, 55000)), 60,
                  "ultralong 60x, 150 kb mean / reads up to 1.5 Mb, 50 kb tandem arrays at 6 copies (BASELINE configs[4]), "
                  "raft -e 60 defaults; reads longer than the LDS window are piled up in pieces"),
    "s50k": (dict(mean_len=20000.0, coverage=30.0), 30,
             "50 k reads, 20 kb mean, 30x (BASELINE configs[1]), raft -e 30 -r 50 -l 20000; launch-bound parity config"),
}
DEFAULT_READS = {"hg002": 3_300_000, "ultralong": 400_000, "s50k": 50_000}


def kernel_source_hash() -> str:
    """Identifies the kernels a counter profile belongs to (profiles/pmc_traffic.json goes stale with them)."""
    h = hashlib.sha1()
    for f in ("pileup_wave.hpp", "wave_launch.hip", "wave_launch.hpp", "pileup_deep.hpp", "pileup.hpp", "engine.hip", "bucket.hpp", "finalize.hpp", "wave.hpp",
              "device_scan.hpp", "pack.hpp", "sort_pairs.hpp", "raft_types.hpp", "engine_ctx.hpp"):
        with open(os.path.join(ROOT, "raft_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    return model, os.cpu_count(), usable


def cpu_baseline(args, o, p):
    """Times the CPU checkers on a bounded sample CUT FROM THE BENCH SET ITSELF: the first `cpu_sample_reads` reads with
    every record whose query is one of them (raft_amd.synth.query_window: a closed problem whose outputs equal the full
    set's for those reads).

    kind "reference": the unmodified reference's own code (create_pileup's bucket fill restated in
    oracle/ref_harness.cpp + repeat_annotate/profileCoverage compiled from /root/reference into
    oracle/_ref/libraft_ref.so in the build container) -- single thread, as the reference ships.
    Falls back to kind "port" (oracle/raft_oracle.c) when the prebuilt reference harness is absent.
    """
    import raft_testlib as tl
    from raft_amd.synth import query_window
    n = min(args.cpu_sample_reads, o.n_reads)
    w = query_window(o, 0, n)
    cols = [c.cpu().numpy() for c in (w.read_len,) + w.columns()]
    t0 = time.perf_counter()
    res = tl.oracle_run(p, *cols)
    t_port = time.perf_counter() - t0
    model, logical, usable = cpu_info()
    out = {"cores": 1, "unit": "PAF records/s", "cpu_model": model, "node_logical_cpus": logical, "node_cpus_usable": usable,
           "sample": f"reads [0, {n}) of the bench set with all {w.n_rec} records whose query is one of them "
                     f"(same generator, same seed {args.seed}), single thread",
           "port_records_per_s": w.n_rec / t_port, "port_seconds": t_port,
           "port_fragments_per_s": len(res["frag_read"]) / t_port}
    if tl.have_ref_lib():
        r = tl.ref_lib_run(p, *cols, want_cov=False)
        t_ref = r["seconds_bucket"] + r["seconds_annotate"]
        out.update(kind="reference", value=w.n_rec / t_ref, reference_seconds=t_ref,
                   note="reference = bucket fill (chop.hpp:155-184 restated) + repeat_annotate() of the unmodified "
                        "reference (profileCoverage + run scan, text streams disabled); break_reads' integer half is "
                        "not separately callable in the reference and is not in this time")
    else:
        out.update(kind="port", value=w.n_rec / t_port,
                   note="oracle/_ref/libraft_ref.so not present (clean checkout without the build container's prebuilt "
                        "reference harness): timing the plain-C restatement instead")
    return out, (w, res)


def windows_of(read_len, reso: int) -> int:
    return int(((read_len.long() + (reso - 1)) // reso).sum())


def grouped_form(torch, hostio, n_reads: int, qid, pinned: bool = False):
    """What the tokeniser knows beside the columns (raft_host_group_offsets): per sorted run, where every read's records
    begin.  Outside every clock, like the tokenisation itself.  Returns an int64 numpy array [n_runs, n_reads + 1]."""
    out = torch.empty(4 * (n_reads + 1), dtype=torch.int64, pin_memory=True).numpy() if pinned else None
    off = hostio.group_offsets(n_reads, qid.cpu().numpy(), max_runs=4, out=out)
    if off is None:
        raise SystemExit("bench.py: the synthetic record stream is not a handful of sorted runs")
    return off


def e2e_leg(args, torch, engine, hostio, o, p, n_iter=3):
    """SURVEY.md §8(d) t_e2e: the tokeniser's arrays in page-locked host memory -> engine -> repeats, fragments and the
    coverage array (transfer encoding: a byte per window + exceptions) back in page-locked host memory.

    The tokeniser hands the engine the symmetric flag (raft_host_paf_symmetric) and the grouped form of the query column
    (raft_host_group_offsets), so what is uploaded is read_len, the per-read record offsets, qs and qe -- 8 bytes per
    record.  Host buffers are allocated (page-locked) before the clock; `first_pass_s` is the first pass of a fresh
    context -- device allocations included -- `seconds` the median of the following ones.  `six_column_input`: the same
    through raft_hip_run_pipelined with the query column (12 bytes per record, rounds 1-2)."""
    import numpy as np
    pe = type(p)(**dict(p.__dict__, symmetric_mode=1))
    host = [c.cpu().pin_memory().numpy() for c in (o.read_len, o.qid, o.qs, o.qe)]
    off = grouped_form(torch, hostio, o.n_reads, o.qid, pinned=True)
    eng = engine.Engine(pe, device=torch.cuda.current_device())
    eng.set_tuning(args.tile_bins, args.force_bucket)
    width = 2 if p.est_cov >= 40 else 1                   # as the CLI chooses: deep sets pile up beyond a byte in repeats
    out = eng.host_output_buffers(host[0], pinned=True, width=width)   # sized by the bounds of include/raft_hip.h, from the read lengths
    out["frag_read"] = torch.empty(out["frag_begin"].size, dtype=torch.int32, pin_memory=True).numpy()
    # (a) chunked, grouped input: upload, pass and download of consecutive read ranges overlap (raft_hip_run_multi_grouped)
    ptimes = []
    for it in range(n_iter + 1):
        t0 = time.perf_counter()
        pres, ps = eng.run_pipelined_grouped(host[0], off, host[2], host[3], out=out)
        ptimes.append(time.perf_counter() - t0)
    psum = (ps.n_bins, ps.n_repeats, ps.n_fragments, ps.total_coverage, ps.total_repeat_length)
    pcopy = {k: pres[k].copy() for k in ("cov8", "rep_s", "rep_e", "frag_begin", "frag_end", "cov_offset", "rep_offset", "frag_offset")}
    # (a') the same with window records: one word per record goes up instead of two (raft_hip_run_multi_windows).  The
    # packing is the tokeniser's (two divisions per record where the coordinates are parsed) and outside the clock like it.
    wrec = None
    if not args.no_windows_leg and p.reso <= 32767:
        wbuf = torch.empty(o.n_rec, dtype=torch.int32, pin_memory=True).numpy().view(np.uint32)
        t0 = time.perf_counter()
        win = hostio.pack_windows(host[2], host[3], p.reso, out=wbuf)
        t_pack = time.perf_counter() - t0
        if win is not None:
            wtimes = []
            for it in range(n_iter + 1):
                t0 = time.perf_counter()
                wres, ws = eng.run_pipelined_windows(host[0], off, win, out=out)
                wtimes.append(time.perf_counter() - t0)
            same_w = psum == (ws.n_bins, ws.n_repeats, ws.n_fragments, ws.total_coverage, ws.total_repeat_length) and \
                all(np.array_equal(pcopy[k], wres[k]) for k in pcopy)
            wsec = sorted(wtimes[1:])[len(wtimes[1:]) // 2]
            wrec = {"records_per_s": o.n_rec / wsec, "fragments_per_s": ws.n_fragments / wsec, "seconds": wsec,
                    "h2d_bytes": host[0].nbytes + off.nbytes + win.nbytes, "equals_coordinate_columns": bool(same_w),
                    "pack_seconds_outside_clock": t_pack,
                    "mode": "raft_hip_run_multi_windows: 4 bytes per record cross PCIe (first window | one past the last << 16, raft_host_pack_windows)"}
            # (a'') ... and the coverage back as four-bit steps (delta4: the pileup's own difference array, large steps and each
            # tile's first window listed with their values, an anchor per 1024 windows): half the download again
            out4 = eng.host_output_buffers(host[0], pinned=True, width=8)
            dtimes = []
            for it in range(n_iter + 1):
                t0 = time.perf_counter()
                dres, ds = eng.run_pipelined_windows(host[0], off, win, out=out4)
                dtimes.append(time.perf_counter() - t0)
            dsec = sorted(dtimes[1:])[len(dtimes[1:]) // 2]
            same_d = psum == (ds.n_bins, ds.n_repeats, ds.n_fragments, ds.total_coverage, ds.total_repeat_length) and \
                all(np.array_equal(pcopy[k], dres[k]) for k in pcopy if k != "cov8")
            # decoded on the host (raft_host_unpack_coverage_d4) and compared with the byte encoding of the same windows, outside the clock
            # (in slices of 2^28 windows -- blocks of 1024 decode independently -- so that the check needs 1 GB, not the 8 of the int32 array)
            lim8 = 255 if width == 1 else 65535

            def d4_equals_bytes(r4, n_bins):
                ok, xi = True, r4["exc_index"]
                for a in range(0, n_bins, 1 << 28):
                    b = min(n_bins, a + (1 << 28))
                    x0, x1 = np.searchsorted(xi, [a, b])
                    dec = hostio.unpack_coverage_d4(b - a, r4["cov_nib"][a // 2:(b + 1) // 2], r4["cov_anchor"][a // 1024:(b + 1023) // 1024],
                                                    xi[x0:x1] - a, r4["exc_value"][x0:x1])
                    ok = ok and bool(np.array_equal(np.minimum(dec, lim8).astype(pcopy["cov8"].dtype), pcopy["cov8"][a:b]))
                    del dec
                return ok
            same_d = same_d and d4_equals_bytes(dres, ds.n_bins)
            wrec["delta4"] = {"records_per_s": o.n_rec / dsec, "fragments_per_s": ds.n_fragments / dsec, "seconds": dsec, "first_pass_s": dtimes[0],
                              "d2h_bytes": int(sum(dres[k].nbytes for k in dres)), "listed_windows": int(dres["exc_index"].size),
                              "decoded_equals_byte_encoding": bool(same_d),
                              "mode": "raft_hip_run_multi_windows, cov_width = RAFT_HIP_COV_DELTA4: coverage comes back as four bits per window"}
            # (a3) SURVEY.md §8(d)'s boundary, nothing prepared: the clock starts with the page-locked int32 columns (read_len, qid, qs, qe)
            # and the tokeniser's symmetric flag, and ends with every output on the host -- ONE call, raft_hip_run_pipelined on the
            # plain columns: the engine's lanes derive every chunk's offsets and window records themselves (engine.hip derive_piece:
            # host threads, into page-locked staging the context keeps) while earlier chunks travel, so 4 bytes per record cross the
            # link and nothing about the stream has to be handed over.
            stimes = []
            for it in range(n_iter + 2):
                t0 = time.perf_counter()
                sres, ss = eng.run_pipelined(host[0], host[1], host[2], host[3], out=out4)
                stimes.append(time.perf_counter() - t0)
            ssec = sorted(stimes[2:])[len(stimes[2:]) // 2]
            # (its chunks are cut elsewhere, so the encoding differs from the prepared-input one where chunks begin: compared decoded)
            same_s = (ss.n_bins, ss.n_repeats, ss.n_fragments, ss.total_coverage, ss.total_repeat_length) == psum and \
                all(np.array_equal(pcopy[k], sres[k]) for k in pcopy if k != "cov8") and d4_equals_bytes(sres, ss.n_bins)
            # ... and what a caller that prepares pays for its first job (the CLI's way: raft_hip_warm_up + raft_hip_reserve beside the
            # tokenising, outside this clock): a fresh context's first call
            e_res = engine.Engine(pe, device=torch.cuda.current_device())
            e_res.set_tuning(args.tile_bins, args.force_bucket)
            e_res.warm_up()
            e_res.reserve(host[0], o.n_rec, 1, 8)
            t0 = time.perf_counter()
            rres, rs = e_res.run_pipelined(host[0], host[1], host[2], host[3], out=out4)
            first_reserved = time.perf_counter() - t0
            same_r = (rs.n_bins, rs.n_repeats, rs.n_fragments, rs.total_coverage, rs.total_repeat_length) == psum
            del rres
            e_res.close()
            # ... and the same boundary with the two derivations as calls of the host library in front of the engine (round 4's first form)
            off_buf = torch.empty(4 * (o.n_reads + 1), dtype=torch.int64, pin_memory=True).numpy()
            xt = []
            for it in range(2):
                t0 = time.perf_counter()
                off_s = hostio.group_offsets(o.n_reads, host[1], max_runs=4, out=off_buf)
                t1 = time.perf_counter()
                win_s = hostio.pack_windows(host[2], host[3], p.reso, out=wbuf)
                t2 = time.perf_counter()
                eng.run_pipelined_windows(host[0], off_s, win_s, out=out4)
                t3 = time.perf_counter()
                xt.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
            xm = min(xt)
            wrec["from_soa"] = {"records_per_s": o.n_rec / ssec, "fragments_per_s": ss.n_fragments / ssec, "seconds": ssec, "first_pass_s": stimes[0],
                                "first_pass_after_reserve_s": first_reserved, "first_pass_after_reserve_equals": bool(same_r),
                                "equals_prepared_input": bool(same_s), "h2d_bytes": host[0].nbytes + 8 * 2 * (o.n_reads + 1) + 4 * o.n_rec,
                                "explicit_host_calls": {"seconds": xm[0], "group_offsets_s": xm[1], "pack_windows_s": xm[2], "engine_s": xm[3]},
                                "boundary": "page-locked int32 columns (read_len, qid, qs, qe) + the tokeniser's symmetric flag in; repeats, fragments and "
                                            "coverage (four-bit steps) in page-locked host memory out; ONE call (raft_hip_run_pipelined): per-read record "
                                            "offsets and window records are derived by the engine's lanes chunk by chunk, inside the clock"}
            del out4, dres, sres
    # (b) chunked, six-column input (query column uploaded, runs guessed from samples, cuts searched)
    ctimes = []
    os.environ["RAFT_NO_DERIVE"] = "1"         # (the columns as they are: what rounds 1-3 uploaded; the derived form is `from_soa` above)
    try:
        for it in range(n_iter):
            t0 = time.perf_counter()
            cres, cs = eng.run_pipelined(host[0], host[1], host[2], host[3], out=out)
            ctimes.append(time.perf_counter() - t0)
    finally:
        del os.environ["RAFT_NO_DERIVE"]
    same_c = psum == (cs.n_bins, cs.n_repeats, cs.n_fragments, cs.total_coverage, cs.total_repeat_length) and \
        all(np.array_equal(pcopy[k], cres[k]) for k in pcopy)
    # (c) one piece: H2D, pass, D2H one after the other (raft_hip_run_host_grouped + raft_hip_fetch_packed)
    times, split = [], None
    eng.set_output_width(width)
    for it in range(n_iter):
        t0 = time.perf_counter()
        eng.run_host_grouped(host[0], off, host[2], host[3])
        s = eng.finish()
        t1 = time.perf_counter()
        got = eng.fetch_packed(out=out)
        t2 = time.perf_counter()
        times.append(t2 - t0)
        split = (t1 - t0, t2 - t1)
    reused = all(got[k].ctypes.data == out[k].ctypes.data for k in got if got[k].size)
    same = psum == (s.n_bins, s.n_repeats, s.n_fragments, s.total_coverage, s.total_repeat_length) and \
        all(np.array_equal(pcopy[k], got[k]) for k in pcopy)
    # the decoded coverage equals what the HBM-resident pass produced (checked on the device, outside the clock)
    limit = 255 if width == 1 else 65535
    dev8 = torch.from_numpy(got["cov8"].astype(np.int32) if width == 2 else got["cov8"]).to(o.read_len.device)
    cov = eng.outputs_device()["cov"]
    ok = bool((dev8.to(torch.int32) == cov.clamp(max=limit)).all()) and int((cov >= limit).sum()) == got["exc_index"].size
    in_bytes = host[0].nbytes + off.nbytes + host[2].nbytes + host[3].nbytes
    out_bytes = sum(a.nbytes for a in got.values())
    eng.close()
    med = lambda v: sorted(v)[len(v) // 2]
    steady, six, one_piece = med(ptimes[1:]), med(ctimes[1:]) if len(ctimes) > 1 else ctimes[0], med(times[1:]) if len(times) > 1 else times[0]
    res = {"records_per_s": o.n_rec / steady, "fragments_per_s": s.n_fragments / steady, "seconds": steady,
            "first_pass_s": ptimes[0],
            "mode": "chunked, grouped input: H2D / pass / D2H of consecutive read ranges overlapped (raft_hip_run_multi_grouped); "
                    "no query column crosses PCIe",
            "six_column_input": {"records_per_s": o.n_rec / six, "seconds": six, "h2d_bytes": sum(a.nbytes for a in host),
                                 "mode": "raft_hip_run_pipelined with RAFT_NO_DERIVE=1 (the three columns uploaded as they are, 12 bytes per record)", "equals_grouped": bool(same_c)},
            "one_piece": {"records_per_s": o.n_rec / one_piece, "seconds": one_piece, "h2d_plus_pass_s": split[0], "pack_plus_d2h_s": split[1]},
            "chunked_equals_one_piece": bool(same),
            "host_memory": "page-locked, allocated before the clock, caller-owned and reused" if reused else "page-locked (grown inside the clock)",
            "h2d_bytes": in_bytes, "d2h_bytes": out_bytes, "coverage_encoding": f"uint{8 * width} per window + (index, value) for windows >= {limit}",
            "exceptions": int(got["exc_index"].size), "symmetric_mode": "asserted by the tokeniser: query sides only",
            "decoded_coverage_equals_device": ok, "passes": n_iter}
    if wrec is not None:
        # the headline of this object is the form the CLI uses (window records); the coordinate-column form stays beside it
        res["coordinate_columns"] = {k: res[k] for k in ("records_per_s", "fragments_per_s", "seconds", "mode", "h2d_bytes")}
        # headline of the object: the contract boundary (from_soa); the prepared-input figures stay beside it
        top = wrec["delta4"] if wrec.get("delta4", {}).get("decoded_equals_byte_encoding") else wrec
        if wrec.get("from_soa", {}).get("equals_prepared_input"):
            res["prepared_input"] = {k: top[k] for k in ("records_per_s", "fragments_per_s", "seconds")}
            res["prepared_input"]["note"] = "offsets and window records built outside the clock (round 3's e2e headline)"
            top = dict(top, **{k: wrec["from_soa"][k] for k in ("records_per_s", "fragments_per_s", "seconds")})
            top["mode"] = "SoA boundary: " + wrec["from_soa"]["boundary"]
        res["byte_per_window_d2h_bytes"] = res["d2h_bytes"]
        if "first_pass_after_reserve_s" in wrec.get("from_soa", {}):
            res["first_pass_after_reserve_s"] = wrec["from_soa"]["first_pass_after_reserve_s"]
        res.update(records_per_s=top["records_per_s"], fragments_per_s=top["fragments_per_s"], seconds=top["seconds"], mode=top["mode"],
                   h2d_bytes=wrec["h2d_bytes"], d2h_bytes=top.get("d2h_bytes", res["d2h_bytes"]), window_records=wrec)
    return res


def spawn_ranks(n: int, limit_s: float) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (no GPU call has been made
    in this process; nothing is ever re-executed) and watch them: the first rank that exits non-zero, or `limit_s` seconds without
    all of them finishing, ends the others (by PID) and the job exits non-zero -- a rank waiting in a collective for a peer that
    has failed would otherwise hang forever."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    t_end = time.monotonic() + limit_s
    rc = 0
    while True:
        codes = [pr.poll() for pr in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() > t_end:
            print(f"bench.py: ranks still running after {limit_s:.0f} s: ending them", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.2)
    for pr in procs:                                     # (the exact processes started above)
        if pr.poll() is None:
            pr.terminate()
    for pr in procs:
        try:
            pr.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="hg002")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's own size); with --strong / --presplit: reads of the ONE set")
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--input", choices=["grouped", "columns", "windows"], default="columns",
                    help="columns (headline): six plain columns, detecting context -- the self-contained pass; grouped: columns + per-read record "
                         "offsets + window count; windows: the same with one word per record (window records, what the CLI hands over)")
    ap.add_argument("--no-qid", action="store_true", help="grouped input without the query column (rebuilt from the offsets on the device)")
    ap.add_argument("--cpu-sample-reads", type=int, default=150_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-windows-leg", action="store_true", help="skip the window-record legs (device pass and host-to-host)")
    ap.add_argument("--cov-width", type=int, default=4, choices=[1, 2, 4, 8],
                    help="bytes per window the timed pass writes (4 = int32 cov[]; 1 / 2 = its transfer encoding; 8 = four-bit steps, delta4)")
    ap.add_argument("--no-packed-leg", action="store_true", help="skip the extra passes that time the pass writing the transfer encoding")
    ap.add_argument("--handover", action="store_true", help="--input columns with symmetric_mode = 1: the symmetric flag is handed over")
    ap.add_argument("--no-six-column-leg", action="store_true", help="skip the extra passes that time the other input form (six-column / grouped)")
    ap.add_argument("--presplit", action="store_true", help="several GPUs (the default): BASELINE configs[3] -- ONE set, the records pre-split across ranks, all-to-all-v in the step")
    ap.add_argument("--host-routed", "--strong", dest="host_routed", action="store_true",
                    help="several GPUs: ONE set cut into --gpus contiguous read ranges, every rank handed the records of its reads (configs[3] without the exchange) as the headline")
    ap.add_argument("--weak", action="store_true", help="several GPUs: every rank an independent set of the workload's size as the headline (rounds 1-5's default)")
    ap.add_argument("--no-extra-legs", "--no-strong-leg", dest="no_extra_legs", action="store_true", help="several GPUs: only the headline form, no legs for the other two")
    ap.add_argument("--presplit-windows", action="store_true", help="(the default since round 6, kept for old command lines)")
    ap.add_argument("--presplit-columns", action="store_true",
                    help="pre-split form: the slices travel as the two coordinate columns (8 bytes per record) instead of window records (4): what reads "
                         "of 65,535 windows or more get in any case")
    ap.add_argument("--watchdog-seconds", type=float, default=1500.0, help="several GPUs: a rank that has not finished by then exits non-zero (0 = off)")
    ap.add_argument("--shuffle", action="store_true", help="the records in random order (create_pileup's bucketing in full: the counting-sort path); never the headline")
    ap.add_argument("--nonsym", action="store_true", help="a non-symmetric PAF (one record per pair: target sides are piled up too, chop.hpp:165-169), shuffled; never the headline")
    ap.add_argument("--tile-bins", type=int, default=0)
    ap.add_argument("--force-bucket", action="store_true")
    ap.add_argument("--plain-input-memory", action="store_true", help="leave the input columns where torch's allocator (hipMalloc) put them instead of moving them "
                    "into memory from raft_hip_device_alloc before the clock (see include/raft_hip.h: a pass's time depends on where its buffers lie)")
    ap.add_argument("--no-placement-ab", action="store_true", help="skip the extra passes that time the pileup kernel with its buffers placed the other ways")
    args = ap.parse_args()
    # several GPUs: which form is the headline (default: configs[3] as written, pre-split)
    args.mode = "weak" if (args.weak or args.gpus <= 1) else ("host_routed" if args.host_routed and not args.presplit else "presplit")
    args.strong = args.mode != "weak"
    args.presplit = args.mode == "presplit"
    if args.shuffle or args.nonsym:                   # (general streams: the plain columns only, no grouped / window / host-to-host legs)
        args.input = "columns"
        args.no_packed_leg = args.no_six_column_leg = args.no_e2e = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, args.watchdog_seconds + 120 if args.watchdog_seconds > 0 else float("inf")))
    if args.gpus > 1 and args.watchdog_seconds > 0:
        # under any launcher: a rank that is still here after the limit (a collective whose peer died, a hung exchange) dumps its
        # threads' stacks and exits non-zero -- the launcher then ends the job instead of waiting forever
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)

    import torch

    from raft_amd import dist as rdist
    from raft_amd import engine, hostio
    from raft_amd.params import RaftParams
    from raft_amd.synth import make_overlaps

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    n_dev = torch.cuda.device_count()
    dist = None
    coll_dev = None
    shared = False
    if world > 1:
        import torch.distributed as dist
        if n_dev >= world:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))     # RCCL over xGMI
            coll_dev = f"cuda:{local}"
        else:
            # fewer GPUs than ranks (a 1-GPU box): ranks share devices and the collectives go through gloo --
            # only good for checking the multi-rank code path, not a scaling number
            local = local % max(n_dev, 1)
            dist.init_process_group("gloo")
            coll_dev = "cpu"
            shared = True
        assert dist.get_world_size() == args.gpus
        if os.environ.get("RAFT_BENCH_FAIL_RANK") == str(rank):      # (tests: a rank that dies before its first collective)
            os._exit(3)
    n_gpus = max(world, 1)
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)

    gen_kw, est_cov, workload_text = WORKLOADS[args.workload]
    if args.shuffle or args.nonsym:
        gen_kw = dict(gen_kw, shuffle=True, symmetric=not args.nonsym)
        workload_text += "; records SHUFFLED" + (", NON-symmetric (one record per pair)" if args.nonsym else "")
    n_reads = args.reads or DEFAULT_READS[args.workload]
    p = RaftParams(est_cov=est_cov)
    p_sym = RaftParams(**dict(p.__dict__, symmetric_mode=1))
    grouped_in = args.input in ("grouped", "windows") and not args.force_bucket
    windows_in = args.input == "windows" and grouped_in

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Where the inputs lie.  They are resident in HBM before any clock starts either way; by default they are moved into device
    # memory from raft_hip_device_alloc first (the engine's placement policy: a virtual range over shuffled 32 MiB chunks) --
    # with torch's hipMalloc blocks the same binary on the same box measured 2.15 .. 2.6 ms for the pileup kernel depending on
    # what earlier processes had left behind (tools/variance_probe.sh, tools/membench).
    mem_ctx = None if args.plain_input_memory else engine.Engine(p, device=local)

    def place(t):
        return t if mem_ctx is None or t is None or not t.is_cuda or t.numel() == 0 else mem_ctx.device_copy(t.contiguous())

    def place_set(ov):
        for name in ("read_len", "qid", "qs", "qe", "tid", "ts", "te"):
            setattr(ov, name, place(getattr(ov, name)))
        return ov

    class Shard:
        """One rank's inputs, resident in HBM: read lengths, the tokeniser's columns and (grouped input) the per-read
        record offsets + the window count."""
        def __init__(self, read_len, cols, want_grouped):
            self.read_len, self.cols = read_len.contiguous(), tuple(c.contiguous() for c in cols)
            self.n_reads, self.n_rec = int(read_len.numel()), int(cols[0].numel())
            self.off = self.n_bins = self.win = None
            if want_grouped:
                self.off = place(torch.as_tensor(grouped_form(torch, hostio, self.n_reads, self.cols[0])).to(dev))
                self.n_bins = windows_of(self.read_len, p.reso)
                if windows_in:
                    w = hostio.pack_windows(self.cols[1].cpu().numpy(), self.cols[2].cpu().numpy(), p.reso)
                    if w is None:
                        raise SystemExit("bench.py: --input windows needs reads below 65,535 windows")
                    self.win = place(torch.as_tensor(w.view("int32")).to(dev))

    def make_engine(sh: Shard, width: int, routed: bool = False):
        # (routed: the records a rank of a host-routed job is handed ARE the query-side multiset of its reads, engine.hip run_routed)
        e = engine.Engine(p_sym if (sh.off is not None or args.handover or routed) else p, device=local)
        e.set_tuning(args.tile_bins, args.force_bucket)
        e.set_output_width(width)
        e.use_torch_stream()
        return e

    def pass_of(e, sh: Shard, qid=True):
        if sh.win is not None:
            e.run_device_windows(sh.read_len, sh.off, sh.win, n_bins=sh.n_bins)
        elif sh.off is not None:
            e.run_device_grouped(sh.read_len, sh.off, sh.cols[0] if qid else None, sh.cols[1], sh.cols[2], n_bins=sh.n_bins)
        else:
            e.run_device(sh.read_len, *sh.cols)
        return e.finish()

    comb = {"work": None, "mine": None, "all": None, "stage": None}

    def combine(s):
        # global read_num base of this shard's fragments + the stdout sums (chop.hpp:195, repeat.hpp:93-97): one all-gather of five
        # integers per rank and step.  It is queued behind the pass and waited for at the next step's (the ranks' passes are
        # independent: nothing of step k+1 needs step k's bases), the last one inside the timed region's closing fence.
        if dist is None:
            return
        if comb["mine"] is None:
            comb["stage"] = torch.zeros(5, dtype=torch.int64).pin_memory() if str(coll_dev) != "cpu" else torch.zeros(5, dtype=torch.int64)
            comb["mine"] = torch.zeros((1, 5), dtype=torch.int64, device=coll_dev)
            comb["all"] = torch.zeros((dist.get_world_size(), 5), dtype=torch.int64, device=coll_dev)
        if comb["work"] is not None:
            comb["work"].wait()
        comb["stage"].copy_(torch.tensor([s.n_fragments, s.total_coverage, s.total_windows, s.total_repeat_length, s.total_read_length], dtype=torch.int64))
        comb["mine"].copy_(comb["stage"].unsqueeze(0))        # (five numbers; the stream is idle behind raft_hip_finish)
        comb["work"] = dist.all_gather_into_tensor(comb["all"], comb["mine"], async_op=True)

    def combine_done():
        if comb["work"] is not None:
            comb["work"].wait()
            comb["work"] = None

    def timed(step_fn, e, warmup, steps):
        for _ in range(warmup):
            s = step_fn()
        pile_t, pass_t = [], []
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            s = step_fn()
            a, b = e.timing()
            pile_t.append(a); pass_t.append(b)
        combine_done()
        fence()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return s, elapsed, sum(pile_t) / len(pile_t), sum(pass_t) / len(pass_t)

    def strong_shard(full, presplit):
        """BASELINE configs[3]: the ONE set cut into `world` contiguous read ranges of near-equal weight.  Host-routed: this
        rank is handed the records whose query is one of its reads (a sub-sequence of every sorted run: still grouped).
        Pre-split: this rank holds the rank-th contiguous slice of the record stream and the step routes it."""
        ipr = torch.bincount(full.qid.long(), minlength=full.n_reads)
        bounds = rdist.partition_reads(full.read_len, p.reso, world, intervals_per_read=ipr)
        b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
        my_len = full.read_len[b0:b1].contiguous()
        if presplit:
            lo, hi = full.n_rec * rank // world, full.n_rec * (rank + 1) // world
            return bounds, my_len, [c[lo:hi].contiguous() for c in full.columns()]
        sel = (full.qid >= b0) & (full.qid < b1)
        cols = [c[sel] for c in full.columns()]
        cols[0] = cols[0] - b0
        return bounds, my_len, cols

    def single_gpu_totals(full):
        e = engine.Engine(p, device=local)
        e.use_torch_stream()
        e.run_device(full.read_len, *full.columns())
        s = e.finish()
        e.close()
        return [s.n_fragments, s.n_repeats, s.total_coverage, s.total_repeat_length, s.n_bins]

    nonlocal_form = ["coordinate columns (8 bytes per record)"]      # what the native exchange of the last pre-split run sent

    def strong_run(full, presplit, warmup, steps):
        """Times the strong-scaling step on this rank's share of `full`; returns (summary, elapsed, kernel s, pass s, records
        of the whole set, totals-equal-single-GPU or None)."""
        bounds, my_len, cols = strong_shard(full, presplit)
        comm = None
        if presplit:
            e = engine.Engine(p_sym, device=local)
            e.set_tuning(args.tile_bins, args.force_bucket)
            e.use_torch_stream()
            if not shared:
                # every rank has a GPU of its own: the native exchange (raft_hip_exchange -- RCCL all-gather of the piece sizes,
                # grouped send / receive of the two coordinate columns and the offset slices; the query ids never travel) feeding
                # the grouped pass.  The communicator's id goes from rank 0 to the others through torch.distributed.
                uid = [engine.Comm.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                comm = engine.Comm(local, uid[0], rank, world)
                off_sl = grouped_form(torch, hostio, full.n_reads, cols[0])                # (the tokeniser's by-product, like the columns)
                my_bins = windows_of(my_len, p.reso)
                bnp = bounds.numpy()
                # the slices as window records (raft_host_pack_windows, the tokeniser's: one word per record): half the bytes over xGMI
                # -- at two ranks a step moves half the set over ONE link -- ; what arrives goes to raft_hip_run_device_windows (more than
                # two runs arriving: unpacked on the device first).  Every rank decides the same way: the longest read of the whole set.
                wv = None
                if not args.presplit_columns and int(full.read_len.max()) < 65535 * p.reso:
                    wv = hostio.pack_windows(cols[1].cpu().numpy(), cols[2].cpu().numpy(), p.reso)
                nonlocal_form[0] = "window records (4 bytes per record)" if wv is not None else "coordinate columns (8 bytes per record)"
                if wv is not None:
                    d_win = torch.as_tensor(wv.view("int32")).to(dev)
                    sl = engine.Slice(off_sl, d_win, None, device_offsets=True)
                else:
                    sl = engine.Slice(off_sl, cols[1], cols[2], device_offsets=True)   # (the offsets stay on the device: nothing is uploaded per step)

                def step():
                    sym = rdist.global_symmetric_flag(cols)                   # broadcast of record 0 + MAX all-reduce
                    assert sym
                    got = comm.exchange(e, bnp, sl)                           # ONE exchange step over xGMI
                    if got["qe"] is None and got["n_rec"]:
                        e.run_device_windows(my_len, got["rec_offset"], got["qs"], n_bins=my_bins)
                    else:
                        e.run_device_grouped(my_len, got["rec_offset"], None, got["qs"], got["qe"], n_bins=my_bins)
                    s = e.finish()
                    combine(s)
                    return s
            else:
                # ranks share a GPU (a one-GPU box: RCCL cannot put two ranks on one device): the exchange in torch over gloo
                def step():
                    cl = [c.cpu() for c in cols]
                    sym = rdist.global_symmetric_flag(cl)                     # broadcast of record 0 + MAX all-reduce
                    iv = rdist.exchange_intervals(cl, bounds, sym)            # one all-to-all-v per column
                    iv = tuple(t.to(dev) for t in iv)
                    s = rdist.run_shard(e, my_len, iv)
                    combine(s)
                    return s
        else:
            sh = Shard(my_len, cols[:3], grouped_in)
            e = make_engine(sh, args.cov_width, routed=True)

            def step():
                s = pass_of(e, sh, qid=not args.no_qid)
                combine(s)
                return s
        s, elapsed, pile, pass_dev = timed(step, e, warmup, steps)
        if comm is not None:
            torch.cuda.synchronize()
            comm.close()
        mine = torch.tensor([s.n_fragments, s.n_repeats, s.total_coverage, s.total_repeat_length, s.n_bins, s.n_intervals], dtype=torch.int64, device=coll_dev or dev)
        if dist is not None:
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        e.close()
        ok = None
        if rank == 0:
            ok = single_gpu_totals(full) == [int(x) for x in mine[:5].tolist()]
        return s, elapsed, pile, pass_dev, [int(x) for x in mine.tolist()], ok

    def weak_run(ov, warmup, steps):
        """Every rank an independent set: one pass over `ov` per step; returns (summary, elapsed, kernel s, pass s, engine, shard)."""
        shw = Shard(ov.read_len, ov.columns(), grouped_in)
        torch.cuda.synchronize()
        ew = make_engine(shw, args.cov_width)

        def step():
            sw = pass_of(ew, shw, qid=not args.no_qid)
            combine(sw)
            return sw
        sw, el, pl, pd = timed(step, ew, warmup, steps)
        return sw, el, pl, pd, ew, shw

    # ---- the timed region
    legs = {}
    if args.strong and world > 1:
        full = place_set(make_overlaps(n_reads, seed=args.seed, device=dev, **gen_kw))
        torch.cuda.synchronize()
        s, elapsed, pile, pass_dev, tot, ok = strong_run(full, args.presplit, args.warmup, args.steps)
        tot_rec, tot_frag, tot_bins, tot_iv = full.n_rec, tot[0], tot[4], tot[5]
        my_rec = s.n_records
        if rank == 0 and not ok:
            raise SystemExit("bench.py: the ranks' totals differ from the single-GPU pass over the same set")
        o, eng, check = full, None, {"ranks_totals_equal_single_gpu_pass": bool(ok)} if rank == 0 else {}
        if not args.no_extra_legs:
            # the other strong form of the SAME set ...
            other = not args.presplit
            s2, el2, pile2, pass2, tot2, ok2 = strong_run(full, other, args.warmup, args.steps)
            if rank == 0:
                legs["presplit" if other else "host_routed"] = {
                    "scaling": "strong", "value": full.n_rec / (el2 / args.steps), "unit": "PAF records/s", "ms_per_step": el2 / args.steps * 1e3,
                    "fragments_per_s": tot2[0] / (el2 / args.steps), "records_total": full.n_rec, "reads_total": full.n_reads,
                    "rank0_reads": s2.n_reads, "rank0_records": s2.n_records, "rank0_kernel_ms": pile2 * 1e3, "rank0_pass_device_ms": pass2 * 1e3,
                    "sharding": f"the ONE set of {full.n_reads} reads in {world} contiguous read ranges, "
                                + ("records pre-split, one all-to-all-v per step" if other else
                                   "host-routed (every rank is handed the records of its reads), no data-path collective"),
                    "ranks_totals_equal_single_gpu_pass": bool(ok2)}
            # ... and the weak form: every rank a set of its own of the workload's size (rank 0's IS the set above)
            ow_ = full if rank == 0 else place_set(make_overlaps(n_reads, seed=args.seed + rank, device=dev, **gen_kw))
            sw, elw, pilew, passw, ew, _ = weak_run(ow_, args.warmup, args.steps)
            cntw = torch.tensor([ow_.n_rec, sw.n_fragments], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(cntw, op=dist.ReduceOp.SUM)
            ew.close()
            if rank == 0:
                legs["weak"] = {"scaling": "weak", "value": int(cntw[0]) / (elw / args.steps), "unit": "PAF records/s", "ms_per_step": elw / args.steps * 1e3,
                                "fragments_per_s": int(cntw[1]) / (elw / args.steps), "records_total": int(cntw[0]), "rank0_kernel_ms": pilew * 1e3,
                                "rank0_pass_device_ms": passw * 1e3,
                                "sharding": f"reads x{world}: every rank an independent set of {n_reads} reads (seed + rank), no data-path collective; "
                                            "NOT BASELINE configs[3] (an N times larger genome): rounds 1-5's multi-GPU headline, kept as a leg"}
            if rank:
                del ow_
    else:
        o = place_set(make_overlaps(n_reads, seed=args.seed + rank, device=dev, **gen_kw))
        s, elapsed, pile, pass_dev, eng, sh = weak_run(o, args.warmup, args.steps)
        my_rec = o.n_rec
        if dist is not None:
            cnt = torch.tensor([my_rec, s.n_fragments, s.n_bins, s.n_intervals], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            tot_rec, tot_frag, tot_bins, tot_iv = (int(x) for x in cnt.tolist())
        else:
            tot_rec, tot_frag, tot_bins, tot_iv = my_rec, s.n_fragments, s.n_bins, s.n_intervals

        # ---- self-check of the last timed pass (outside the clock): size-independent invariants of the outputs
        out = eng.outputs_device()
        touched = ((o.qe.long() - 1) // p.reso - o.qs.long() // p.reso + 1).clamp(min=0)
        if args.nonsym:                               # (target sides of records whose two reads differ are piled up as well)
            touched = torch.cat([touched, (((o.te.long() - 1) // p.reso - o.ts.long() // p.reso + 1).clamp(min=0))[o.tid != o.qid]])
        fo, fb, fe = out["frag_offset"], out["frag_begin"], out["frag_end"]
        same = out["frag_read"][1:] == out["frag_read"][:-1]
        check = {"sum_cov_equals_windows_touched": int(out["cov"].sum(dtype=torch.int64)) == int(touched.sum()) == s.total_coverage,
                 "fragments_tile_reads": bool((fb[fo[:-1]] == 0).all()) and bool((fe[fo[1:] - 1] == o.read_len).all())
                 and bool(((fe[:-1] - fb[1:])[same] == p.overlap_length).all()),
                 "windows": s.n_bins == int(((o.read_len.long() + p.reso - 1) // p.reso).sum()),
                 # cut points (written by the pass): every read's first is 0 and its last its length, ascending in between, and every
                 # fragment boundary is one of them (chop.hpp:225-246, :280-321)
                 "cut_points_span_reads": bool((out["cuts"][out["cut_offset"][:-1]] == 0).all()) and bool((out["cuts"][out["cut_offset"][1:] - 1] == o.read_len).all())
                 and int(out["cut_offset"][-1]) == s.n_cuts}
        del touched, out
        if not all(check.values()) and "RAFT_BENCH_ABLATION" not in os.environ:      # (diagnostic builds with parts of the kernel switched off compute nonsense on purpose)
            raise SystemExit(f"bench.py: self-check failed: {check}")

        # ---- several GPUs, weak headline (--weak): BASELINE configs[3]'s setting beside it -- the ONE set of configs[2] in `world` read ranges
        if world > 1 and not args.no_extra_legs:
            full = place_set(make_overlaps(n_reads, seed=args.seed, device=dev, **gen_kw)) if rank else o   # (rank 0's weak shard IS that set)
            torch.cuda.synchronize()
            for form, pre in (("presplit", True), ("host_routed", False)):
                s2, el2, pile2, pass2, tot2, ok2 = strong_run(full, pre, args.warmup, args.steps)
                if rank == 0:
                    legs[form] = {"scaling": "strong", "value": full.n_rec / (el2 / args.steps), "unit": "PAF records/s", "ms_per_step": el2 / args.steps * 1e3,
                                  "fragments_per_s": tot2[0] / (el2 / args.steps), "records_total": full.n_rec, "reads_total": full.n_reads,
                                  "rank0_reads": s2.n_reads, "rank0_records": s2.n_records, "rank0_kernel_ms": pile2 * 1e3, "rank0_pass_device_ms": pass2 * 1e3,
                                  "sharding": f"the ONE set of {full.n_reads} reads in {world} contiguous read ranges, "
                                              + ("records pre-split, one all-to-all-v per step" if pre else
                                                 "host-routed (every rank is handed the records of its reads), no data-path collective"),
                                  "ranks_totals_equal_single_gpu_pass": bool(ok2)}
            if rank:
                del full

    # ---- what the placement policy buys on THIS box: the headline pass once more in this process with
    # every buffer -- the engine's and the input columns -- made under each policy: eight-fold spread chunks (the default),
    # chunks taken one after the other, plain hipMalloc.  Outside the headline's clock; kernel and pass by HIP events.
    placement_ab = None
    if n_gpus == 1 and not args.no_placement_ab and not (args.strong and world > 1) and args.input == "columns" and not args.shuffle and not args.nonsym:
        placement_ab = {}
        before = engine.set_placement(8)
        for name, spread in (("spread8", 8), ("spread1", 1), ("hipMalloc", 0)):
            engine.set_placement(spread)
            ea = engine.Engine(p, device=local)
            ea.set_tuning(args.tile_bins, args.force_bucket)
            ea.use_torch_stream()
            colsa = [ea.device_copy(c) for c in (o.read_len,) + tuple(o.columns())]
            kt, pt = [], []
            for it in range(7):
                ea.run_device(*colsa)
                sa = ea.finish()
                if it >= 2:
                    ka, pa_ = ea.timing()
                    kt.append(ka * 1e3); pt.append(pa_ * 1e3)
            placement_ab[name] = {"kernel_ms": sum(kt) / len(kt), "pass_device_ms": sum(pt) / len(pt), "fragments": int(sa.n_fragments)}
            del colsa
            ea.close()
        engine.set_placement(before)
        placement_ab["note"] = ("pileup kernel / whole pass, mean of 5 passes after 2, one process, buffers AND input columns made under the policy; "
                                "the headline runs under the library's default (spread8 unless RAFT_VMM_SPREAD / RAFT_NO_VMM say otherwise)")

    # ---- the pass exactly as the CLI and the host pipelines run it: grouped input WITHOUT the query column (it never crosses
    # PCIe: rebuilt from the offsets on the device), the pileup kernel writing the transfer encoding of cov[] (one byte per
    # window, two from -e 40 on, + the windows at or above the limit).  Checked against the int32 pass (outside the clock).
    packed = windows_leg = windows_d4_leg = None
    if n_gpus == 1 and not args.no_packed_leg and args.cov_width == 4:
        w = 2 if p.est_cov >= 40 else 1
        shp = sh if sh.off is not None else Shard(o.read_len, o.columns(), True)
        ref_cov = eng.outputs_device()["cov"]
        lim = 255 if w == 1 else 65535
        big = (ref_cov >= lim).nonzero().flatten()

        def encoded_pass(form):
            d_win = None
            d4 = form == "windows_d4"
            if form in ("windows", "windows_d4"):
                win = hostio.pack_windows(shp.cols[1].cpu().numpy(), shp.cols[2].cpu().numpy(), p.reso)   # (the tokeniser's, outside every clock)
                if win is None:
                    return None
                d_win = place(torch.as_tensor(win.view("int32")).to(dev))
            e3 = engine.Engine(p_sym, device=local)
            e3.set_tuning(args.tile_bins, args.force_bucket)
            e3.set_output_width(8 if d4 else w)
            e3.use_torch_stream()
            kt, pt, wall = [], [], []
            for it in range(6):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                if d_win is None:
                    s3 = pass_of(e3, shp, qid=False)
                else:
                    e3.run_device_windows(shp.read_len, shp.off, d_win, n_bins=shp.n_bins)
                    s3 = e3.finish()
                torch.cuda.synchronize()
                if it:
                    wall.append(time.perf_counter() - t1)
                    a, b = e3.timing(); kt.append(a); pt.append(b)
            pk = e3.packed_device()
            assert pk is not None and pk["width"] == (8 if d4 else w)
            if d4:       # decoded on the device (outputs_device) against the int32 pass; the listed windows carry their values
                n_listed = int(pk["exc_index"].numel())
                ok = bool(torch.equal(pk["exc_value"], ref_cov[pk["exc_index"]]))
                ok = ok and bool(torch.equal(e3.outputs_device()["cov"], ref_cov))
                codes = order = None
            else:
                codes = pk["cov8"] if w == 1 else pk["cov8"].to(torch.int32) & 0xFFFF
                ok = bool((codes == ref_cov.clamp(max=lim)).all())
                order = pk["exc_index"].argsort()
                ok = ok and bool(torch.equal(pk["exc_index"][order], big)) and bool(torch.equal(pk["exc_value"][order], ref_cov[big]))
                n_listed = int(pk["exc_index"].numel())
            ok = ok and (s3.n_fragments, s3.n_repeats, s3.total_coverage, s3.total_repeat_length) == (s.n_fragments, s.n_repeats, s.total_coverage, s.total_repeat_length)
            if not ok:
                raise SystemExit(f"bench.py: the packed-output pass ({form}) differs from the int32 pass")
            n_runs = shp.off.shape[0]
            if d_win is None:    # 8 B per record read (qs, qe; the id comes from 8 B per read and run, read twice: expanded, then by the cuts), w per window written
                bytes_p = 8 * s3.n_intervals + 16 * s3.n_reads * n_runs
                text = "grouped, no query column (ids rebuilt from the offsets on the device)"
            else:                # 4 B per record read; 8 B per read and run (the tile's slice of the offsets)
                bytes_p = 4 * s3.n_intervals + 8 * s3.n_reads * n_runs
                text = "grouped, window records (one word per record: first window | one past the last << 16; raft_hip_run_device_windows)"
            if d4:
                text = text.replace("raft_hip_run_device_windows)", "raft_hip_run_device_windows), coverage written as four-bit steps (delta4)")
                bytes_p += s3.n_bins // 2 + 4 * (s3.n_bins // 1024) + 12 * n_listed + 4 * s3.n_reads + 8 * s3.n_repeats
            else:
                bytes_p += w * s3.n_bins + 12 * n_listed + 4 * s3.n_reads + 8 * s3.n_repeats
            k_s, p_s, w_s = sum(kt) / len(kt), sum(pt) / len(pt), sum(wall) / len(wall)
            res = {"cov_width": 8 if d4 else w, "input": text, "value": my_rec / w_s, "unit": "PAF records/s",
                   "ms_per_step": w_s * 1e3, "kernel_ms": k_s * 1e3, "pass_device_ms": p_s * 1e3, "bytes_algorithmic": bytes_p,
                   "kernel_frac": bytes_p / k_s / 1e9 / HBM_PEAK_GBS, "pass_frac": bytes_p / p_s / 1e9 / HBM_PEAK_GBS,
                   "n_exceptions": n_listed, "equals_int32_pass": ok}
            del codes, order, pk, d_win
            e3.close()
            return res
        packed = encoded_pass("columns")
        if not args.no_windows_leg and p.reso <= 32767:
            windows_leg = encoded_pass("windows")
            if windows_leg is not None:
                windows_d4_leg = encoded_pass("windows_d4")
        del ref_cov, big

    # ---- the other input form beside the headline: the six plain columns into a detecting context (with its inspect-first
    # form) when the headline is a grouped form; the grouped form (offsets + window count handed over) when the headline is
    # the six-column pass.  Same set, same checks, five passes each.
    six = grouped_leg = no_cuts = None

    def extra_passes(form, env=None, cuts=True):
        if env:
            os.environ[env] = "1"
        try:
            if form == "grouped":
                shg = sh if sh.off is not None else Shard(o.read_len, o.columns(), True)
                e2 = engine.Engine(p_sym, device=local)
            else:
                shg = None
                e2 = engine.Engine(p_sym if form == "handover" else p, device=local)
            e2.set_tuning(args.tile_bins, args.force_bucket)
            e2.set_emit_cuts(cuts)
            e2.use_torch_stream()
            kt, pt, wall = [], [], []
            for it in range(5):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                if shg is not None:
                    e2.run_device_grouped(shg.read_len, shg.off, shg.cols[0], shg.cols[1], shg.cols[2], n_bins=shg.n_bins)
                elif form == "handover":
                    e2.run_device(o.read_len, o.qid, o.qs, o.qe)
                else:
                    e2.run_device(o.read_len, *o.columns())
                s2 = e2.finish()
                torch.cuda.synchronize()
                if it:
                    wall.append(time.perf_counter() - t1)
                    a, b = e2.timing(); kt.append(a); pt.append(b)
            assert (s2.n_fragments, s2.n_repeats, s2.total_coverage, s2.symmetric) == (s.n_fragments, s.n_repeats, s.total_coverage, 1)
            e2.close()
            return sum(kt) / len(kt), sum(pt) / len(pt), sum(wall) / len(wall)
        finally:
            if env:
                del os.environ[env]
    if n_gpus == 1 and not args.no_six_column_leg and eng is not None and args.cov_width == 4 and not args.force_bucket:
        if grouped_in:
            k2, p2, w2 = extra_passes("columns")
            _, p2i, _ = extra_passes("columns", "RAFT_ALWAYS_INSPECT")
            six = {"value": my_rec / w2, "unit": "PAF records/s", "ms_per_step": w2 * 1e3, "kernel_ms": k2 * 1e3, "pass_device_ms": p2 * 1e3,
                   "pass_device_ms_inspect_first": p2i * 1e3,
                   "form": "raft_hip_run_device, symmetric_mode = -1: sorted runs guessed from samples, tile cuts searched, mirror of record 0 found beside the pileup, one host wait"}
        else:
            k2, p2, w2 = extra_passes("grouped")
            grouped_leg = {"value": my_rec / w2, "unit": "PAF records/s", "ms_per_step": w2 * 1e3, "kernel_ms": k2 * 1e3, "pass_device_ms": p2 * 1e3,
                           "form": "raft_hip_run_device_grouped, symmetric_mode = 1: the caller hands over the symmetric flag, per sorted run where every read's records "
                                   "begin (raft_host_group_offsets, built OUTSIDE this clock) and the window count: no guess, no searches, no host wait",
                           "note": "part of create_pileup's bucketing is done by the caller here: never the headline"}
            _, p2i, _ = extra_passes("columns", "RAFT_ALWAYS_INSPECT")
            grouped_leg["six_column_pass_device_ms_inspect_first"] = p2i * 1e3
        # the headline's form with the cut points left to the first fetch (what rounds 1-3 timed)
        form0 = "grouped" if grouped_in else ("handover" if args.handover else "columns")
        if not windows_in:
            _, p0, w0 = extra_passes(form0, cuts=False)
            no_cuts = {"pass_device_ms": p0 * 1e3, "ms_per_step": w0 * 1e3}

    if rank == 0:
        per_step = elapsed / args.steps
        # dominant kernel: pileup + prefix scan + coverage store + run scan (pileup_wave.hpp).
        # algorithmic bytes per launch (this rank): 12 B per interval read once, 4 B per window written once,
        # 4 B per read (length) and 8 B per repeat emitted
        cov_bytes = s.n_bins // 2 + 4 * (s.n_bins // 1024) if args.cov_width == 8 else args.cov_width * s.n_bins
        bytes_alg = (4 if windows_in else 12) * s.n_intervals + cov_bytes + 4 * s.n_reads + 8 * s.n_repeats
        # what the pileup KERNEL reads is not always what the pass is handed: behind the general bucketing (a stream that is not a
        # handful of sorted runs) the sort leaves one 4-byte window record per interval plus 8 bytes per read of offsets -- the kernel's
        # fraction is priced on that (12 bytes per interval there printed 0.83, above what a plain copy reaches), the
        # pass's fraction on the algorithmic bytes of the whole pass as before
        bucket_windows = s.interval_path == 1 and bool(getattr(s, "flags", 0) & 1)
        bytes_kernel = bytes_alg if not bucket_windows else 4 * s.n_intervals + 8 * s.n_reads + cov_bytes + 4 * s.n_reads + 8 * s.n_repeats
        achieved = bytes_kernel / pile / 1e9
        strong_line = args.strong and world > 1
        if args.presplit and world > 1:
            input_text = "records pre-split across ranks; the received intervals enter as query-side records (raft_hip_run_device, symmetric_mode = 1)"
        elif windows_in:
            input_text = "grouped, window records (raft_hip_run_device_windows): one word per record, read ids from the per-read record offsets"
        elif grouped_in:
            input_text = ("grouped (raft_hip_run_device_grouped): columns + per-read record offsets of every sorted run + window count, as the CLI hands them over"
                          + ("; no query column" if args.no_qid else "; every record checked against its tile's reads"))
        else:
            input_text = "six plain columns (raft_hip_run_device)" + (", symmetric flag handed over" if args.handover else ", detecting context")
        line = {
            "metric": "PAF overlap records/s + fragments/s, 32x human all-vs-all; HBM GB/s vs peak",
            "value": tot_rec / per_step, "unit": "PAF records/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "strong" if strong_line else "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "fragments_per_s": tot_frag / per_step,
            "config": {"workload": workload_text, "workload_name": args.workload, "input": input_text,
                       "input_memory": "torch allocator (hipMalloc)" if mem_ctx is None else "raft_hip_device_alloc (device memory placed as the engine places its own arrays: shuffled 32 MiB chunks), filled before the clock",
                       "reads_per_gpu": s.n_reads, "records_per_gpu": my_rec, "records_total": tot_rec,
                       "windows_total": tot_bins, "intervals_total": tot_iv, "fragments_total": tot_frag,
                       "repeats_rank0": s.n_repeats, "mean_read_len": gen_kw["mean_len"], "coverage": gen_kw["coverage"],
                       "interval_path": "sorted-segments" if s.interval_path == 0 else "counting-sort",
                       "segments": s.n_segments,
                       "cut_points": "written by the pass (chop.hpp:225-246 final_stars; finalize_fill_kernel<true>), inside the step and the clock",
                       "cut_points_total_rank0": s.n_cuts,
                       "sharding": (f"the ONE set of {o.n_reads} reads in {n_gpus} contiguous read ranges, "
                                    + ("records pre-split, one all-to-all-v per step" if args.presplit else "host-routed, no data-path collective") if strong_line
                                    else f"reads x{n_gpus}, independent shards, no data-path collective") + (", ranks share GPUs (gloo check run)" if shared else ""),
                       **({"exchange_form": nonlocal_form[0]} if (strong_line and args.presplit and world > 1 and not shared) else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "pileup_wave_kernel",
                         "kernel_ms": pile * 1e3, "bytes_algorithmic": bytes_alg, "bytes_kernel": bytes_kernel,
                         "bytes_kernel_note": ("the pileup kernel reads the sorted window records the bucketing left (4 B per interval + 8 B per read of offsets), not the "
                                               "12 B per interval the pass was handed: `frac` is priced on that, `pass_frac` on the pass's algorithmic bytes") if bucket_windows
                                              else "the kernel reads what the pass is handed: the algorithmic bytes",
                         "pass_device_ms": pass_dev * 1e3, "pass_achieved": bytes_alg / pass_dev / 1e9,
                         "pass_frac": bytes_alg / pass_dev / 1e9 / HBM_PEAK_GBS, "pass_includes_cut_points": True,
                         "pass_note": "pass_frac divides the pileup kernel's algorithmic bytes (12 I + 4 B + 4 N + 8 R) by the device time of the whole pass, "
                                      "cut points (4 M more bytes, not counted) included",
                         "kernel_source_hash": kernel_source_hash()},
            "self_check": check,
        }
        if packed is not None:
            # the product path (what the CLI runs) next to the int32 figures: a byte per window written, no query column read
            line["roofline"].update(product_path_kernel_ms=packed["kernel_ms"], product_path_frac=packed["kernel_frac"],
                                    product_path_pass_device_ms=packed["pass_device_ms"], product_path_pass_frac=packed["pass_frac"],
                                    product_path_bytes_algorithmic=packed["bytes_algorithmic"])
            line["packed_output"] = packed
        if windows_leg is not None:
            # ... and with the records as the tokeniser can hand them over: one word each (the CLI's form)
            line["roofline"].update(product_path_kernel_ms=windows_leg["kernel_ms"], product_path_frac=windows_leg["kernel_frac"],
                                    product_path_pass_device_ms=windows_leg["pass_device_ms"], product_path_pass_frac=windows_leg["pass_frac"],
                                    product_path_bytes_algorithmic=windows_leg["bytes_algorithmic"], product_path_input="window records")
            line["window_records"] = windows_leg
            if windows_d4_leg is not None:
                line["window_records_delta4"] = windows_d4_leg
        if six is not None:
            line["six_column"] = six
        if grouped_leg is not None:
            line["grouped"] = grouped_leg
        if no_cuts is not None:
            line["roofline"]["pass_device_ms_without_cuts"] = no_cuts["pass_device_ms"]
            line["roofline"]["ms_per_step_without_cuts"] = no_cuts["ms_per_step"]
        for name, leg in legs.items():
            line[name] = leg
        if args.cov_width != 4:
            line["config"]["cov_width"] = args.cov_width
        # (per workload and input form; the headline workload's: pmc_traffic.json, with window records in / a byte per window out: pmc_traffic_windows_w1.json)
        form = (f"_windows_w{args.cov_width}" if windows_in else ("" if args.cov_width == 4 else f"_w{args.cov_width}"))
        traffic_file = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}{form}.json")
        if not os.path.exists(traffic_file):
            traffic_file = os.path.join(ROOT, "profiles", f"pmc_traffic{form}.json")
        if os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                # counters belong to one workload AND one build of the kernels: anything else is stale and stays null
                if tj.get("records_per_gpu") == my_rec and tj.get("kernel_source_hash") == line["roofline"]["kernel_source_hash"]:
                    line["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = tj.get("source")
                else:
                    line["roofline"]["traffic_note"] = "profiles/pmc_traffic.json is from another workload or kernel build (stale): not used"
            except Exception:
                pass
        if n_gpus == 1:
            if not args.no_e2e:
                try:
                    line["e2e"] = e2e_leg(args, torch, engine, hostio, o, p)
                except engine.RaftError as ex:
                    line["e2e"] = {"error": str(ex), "note": "transfer encoding of cov[] not suited to this coverage: the caller has to "
                                   "provide room for one exception per window at or above the limit"}
            if not args.no_cpu_baseline:
                cb, _ = cpu_baseline(args, o, p)
                line["cpu_baseline"] = cb
        if placement_ab is not None:
            line["placement_ab"] = placement_ab
        if eng is not None:
            tr = eng.placement_trial()
            if tr is not None:
                # the headline context's own choice (its first, untimed pass): the pileup kernel into the coverage array as the policy placed
                # it and into a few more candidates (plain blocks, other chunk mappings); the timed passes run with the one that was kept
                line["placement_trial"] = {"first_placement_ms": tr[0], "best_other_candidate_ms": tr[1],
                                           "kept": ("first placement", "a plain hipMalloc block", "another chunk mapping")[tr[2]],
                                           "note": "opt-in (RAFT_PLACEMENT_TRIALS=<k>): the context's first (untimed) pass ran the pileup kernel into k candidate coverage arrays, warm; the timed passes use the fastest"}
        print(json.dumps(line))
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
