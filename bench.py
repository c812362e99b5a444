#!/usr/bin/env python3
"""bench.py -- RAFT hot path (PAF overlaps -> coverage -> repeat mask -> fragments) on MI355X.

One "step" = one full pass of the engine (raft_hip_run_device + raft_hip_finish) over one
synthetic all-vs-all overlap set that is already resident in HBM when the clock starts:
record inspection, per-tile interval ranges (or counting-sort bucketing), the pileup /
prefix-sum / run-scan kernel, repeat ordering, cut points, fragment table and the stdout
statistics.  Outputs stay in HBM.

Workload (config.workload), default `hg002`: BASELINE.json configs[2] restated synthetically (SURVEY.md §8d,
config 3): HG002-like 32x set, 3.3 M reads of 30 kb mean length, ~2.9e8 symmetric PAF records
written as a cis file followed by a trans file, each grouped by ascending query id.  With
--gpus N every rank owns an independent shard of that size (reads and their overlaps shard
embarrassingly; no data-path collective) -- weak scaling; the only collective is the
all-gather of per-rank fragment totals that turns local fragment ids into global read_num.
Other workloads (never the headline line): `ultralong` = configs[4] (60x, 150 kb mean, reads up to 1.5 Mb, 50 kb
tandem arrays), `s50k` = configs[1] (50 k reads, 20 kb, 30x).

The line also carries `e2e` (N = 1): the same workload from page-locked host columns to every output back on the
host (SURVEY.md §8d `t_e2e`; never the headline `value`), and `roofline.pass_frac`: algorithmic bytes over the device
time of the WHOLE pass, not only the dominant kernel.

`python bench.py --gpus N` without a launcher starts the N ranks itself (child processes, before any GPU call);
under `python -m torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE.  `--presplit` puts BASELINE
configs[3]'s exchange into the timed step: every rank holds a contiguous slice of the record stream and one
all-to-all-v (RCCL over xGMI) routes the intervals to the ranks that own their reads.

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
                  "raft -e 60 defaults; reads longer than the LDS window take the chunked general kernel"),
    "s50k": (dict(mean_len=20000.0, coverage=30.0), 30,
             "50 k reads, 20 kb mean, 30x (BASELINE configs[1]), raft -e 30 -r 50 -l 20000; launch-bound parity config"),
}
DEFAULT_READS = {"hg002": 3_300_000, "ultralong": 400_000, "s50k": 50_000}


def kernel_source_hash() -> str:
    """Identifies the kernels a counter profile belongs to (profiles/pmc_traffic.json goes stale with them)."""
    h = hashlib.sha1()
    for f in ("pileup_fast.hpp", "pileup.hpp", "engine.hip", "bucket.hpp", "finalize.hpp", "wave.hpp", "device_scan.hpp"):
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


def e2e_leg(args, torch, engine, o, p, n_iter=3):
    """SURVEY.md §8(d) t_e2e: int32 SoA columns in page-locked host memory -> engine -> repeats, fragments and the
    coverage array (transfer encoding: a byte per window + exceptions) back in page-locked host memory.

    The tokeniser hands the engine the symmetric flag (raft_host_paf_symmetric), so symmetric_mode = 1 and only the
    three query columns are uploaded.  Host buffers are allocated (page-locked) before the clock; `first_pass_s` is the
    first pass of a fresh context -- device allocations included -- `seconds` the median of the following ones."""
    import numpy as np
    assert p.symmetric_mode == -1
    pe = type(p)(**dict(p.__dict__, symmetric_mode=1))
    host = [c.cpu().pin_memory().numpy() for c in (o.read_len, o.qid, o.qs, o.qe)]
    eng = engine.Engine(pe, device=torch.cuda.current_device())
    eng.set_tuning(args.tile_bins, args.force_bucket, args.variant)
    width = 2 if p.est_cov >= 40 else 1                   # as the CLI chooses: deep sets pile up beyond a byte in repeats
    out = eng.host_output_buffers(host[0], pinned=True, width=width)   # sized by the bounds of include/raft_hip.h, from the read lengths
    out["frag_read"] = torch.empty(out["frag_begin"].size, dtype=torch.int32, pin_memory=True).numpy()
    # (a) chunked: upload, pass and download of consecutive read ranges overlap (raft_hip_run_pipelined)
    ptimes = []
    for it in range(n_iter + 1):
        t0 = time.perf_counter()
        pres, ps = eng.run_pipelined(host[0], host[1], host[2], host[3], out=out)
        ptimes.append(time.perf_counter() - t0)
    psum = (ps.n_bins, ps.n_repeats, ps.n_fragments, ps.total_coverage, ps.total_repeat_length)
    pcopy = {k: pres[k].copy() for k in ("cov8", "rep_s", "rep_e", "frag_begin", "frag_end", "cov_offset", "rep_offset", "frag_offset")}
    # (b) one piece: H2D, pass, pack, D2H one after the other (raft_hip_run_host + raft_hip_fetch_packed)
    times, split = [], None
    for it in range(n_iter + 1):
        t0 = time.perf_counter()
        eng.run_host(host[0], host[1], host[2], host[3], None, None, None)
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
    in_bytes = sum(a.nbytes for a in host)
    out_bytes = sum(a.nbytes for a in got.values())
    eng.close()
    steady = sorted(ptimes[1:])[len(ptimes[1:]) // 2]
    one_piece = sorted(times[1:])[len(times[1:]) // 2]
    return {"records_per_s": o.n_rec / steady, "fragments_per_s": s.n_fragments / steady, "seconds": steady,
            "first_pass_s": ptimes[0], "mode": "chunked: H2D / pass / D2H of consecutive read ranges overlapped (raft_hip_run_pipelined)",
            "one_piece": {"records_per_s": o.n_rec / one_piece, "seconds": one_piece, "h2d_plus_pass_s": split[0], "pack_plus_d2h_s": split[1]},
            "chunked_equals_one_piece": bool(same),
            "host_memory": "page-locked, allocated before the clock, caller-owned and reused" if reused else "page-locked (grown inside the clock)",
            "h2d_bytes": in_bytes, "d2h_bytes": out_bytes, "coverage_encoding": f"uint{8 * width} per window + (index, value) for windows >= {limit}",
            "exceptions": int(got["exc_index"].size), "symmetric_mode": "asserted by the tokeniser: query columns only",
            "decoded_coverage_equals_device": ok, "passes": n_iter}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (no GPU call has been made
    in this process) and return the first non-zero exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        code = pr.wait()
        rc = rc or code
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="hg002")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's own size)")
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--cpu-sample-reads", type=int, default=150_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--cov-width", type=int, default=4, choices=[1, 2, 4], help="bytes per window the timed pass writes (4 = int32 cov[]; 1 / 2 = its transfer encoding)")
    ap.add_argument("--no-packed-leg", action="store_true", help="skip the extra passes that time the pass writing the transfer encoding")
    ap.add_argument("--handover", action="store_true", help="symmetric_mode = 1: the symmetric flag is handed over, as the CLI does")
    ap.add_argument("--no-detect-leg", action="store_true", help="skip the extra passes that time the inspect-first form")
    ap.add_argument("--presplit", action="store_true", help="BASELINE configs[3]: records pre-split across ranks, all-to-all-v in the step")
    ap.add_argument("--tile-bins", type=int, default=0)
    ap.add_argument("--force-bucket", action="store_true")
    ap.add_argument("--variant", type=int, default=-1, help="pileup kernel variant (engine.hip kVariants), -1 = default")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch

    from raft_amd import dist as rdist
    from raft_amd import engine
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
    n_gpus = max(world, 1)
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)

    gen_kw, est_cov, workload_text = WORKLOADS[args.workload]
    n_reads = args.reads or DEFAULT_READS[args.workload]
    p = RaftParams(est_cov=est_cov)

    # ---- synthetic shard, generated on the device (resident in HBM before the clock starts)
    if not args.presplit or world == 1:
        o = make_overlaps(n_reads, seed=args.seed + rank, device=dev, **gen_kw)
        cols = (o.read_len,) + o.columns()
        my_rec = o.n_rec
    else:
        # BASELINE configs[3]: ONE global record stream, rank g holds its g-th contiguous slice.  The stream is built so
        # that every slice holds records of every rank's reads: set h (seed + h, reads [h*n, (h+1)*n) globally) is
        # generated in turn and its g-th n-th part is appended to rank g's slice.
        parts, lens = [], []
        for h in range(world):
            oh = make_overlaps(n_reads, seed=args.seed + h, device=dev, **gen_kw)
            lo, hi = oh.n_rec * rank // world, oh.n_rec * (rank + 1) // world
            q, qs, qe, t, ts, te = (c[lo:hi] for c in oh.columns())
            parts.append((q + h * n_reads, qs, qe, t + h * n_reads, ts, te))
            lens.append(oh.read_len)
            del oh
        slice_cols = [torch.cat([pp[k] for pp in parts]).contiguous() for k in range(6)]
        read_len_all = torch.cat(lens)
        del parts, lens
        my_rec = int(slice_cols[0].numel())
        bounds = rdist.partition_reads(read_len_all, p.reso, world)
        b0, b1 = int(bounds[rank]), int(bounds[rank + 1])
        my_len = read_len_all[b0:b1].contiguous()
    torch.cuda.synchronize()

    # The engine is self-contained here (symmetric_mode = -1: it finds out by itself that the PAF is symmetric).  Its pass is
    # built on a sampled guess of the sorted runs and the assumption of a symmetric PAF, both verified while it runs (the
    # runs in the pileup kernels, the mirror of record 0 by a one-workgroup kernel beside them); `--handover` runs it as
    # the `raft` CLI does, with the flag the tokeniser found (symmetric_mode = 1: no target columns needed at all).  The
    # default line also times the form that looks at every record first (inspect pass): roofline.pass_device_ms_inspect_first.
    p_run = RaftParams(**dict(p.__dict__, symmetric_mode=1)) if (args.handover or (args.presplit and world > 1)) else p
    eng = engine.Engine(p_run, device=local)
    eng.set_tuning(args.tile_bins, args.force_bucket, args.variant)
    eng.set_output_width(args.cov_width)
    eng.use_torch_stream()

    def step():
        if args.presplit and world > 1:
            cl = [c if not shared else c.cpu() for c in slice_cols]
            sym = rdist.global_symmetric_flag(cl)                         # broadcast of record 0 + MAX all-reduce
            iv = rdist.exchange_intervals(cl, bounds, sym)                # ONE all-to-all-v (RCCL over xGMI)
            iv = tuple(t.to(dev) for t in iv)
            s = rdist.run_shard(eng, my_len, iv)
        else:
            eng.run_device(*cols)
            s = eng.finish()
        if dist is not None:  # global read_num base of this shard's fragments + the stdout sums (chop.hpp:195, repeat.hpp:93-97)
            rdist.combine_totals(s.n_fragments, s.total_coverage, s.total_windows, s.total_repeat_length,
                                 s.total_read_length, device=coll_dev)
        return s

    for _ in range(args.warmup):
        s = step()
    pile_t, pass_t = [], []

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s = step()
        a, b = eng.timing()
        pile_t.append(a); pass_t.append(b)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([my_rec, s.n_fragments, s.n_bins, s.n_intervals], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_rec, tot_frag, tot_bins, tot_iv = (int(x) for x in cnt.tolist())
    else:
        tot_rec, tot_frag, tot_bins, tot_iv = my_rec, s.n_fragments, s.n_bins, s.n_intervals

    # ---- self-check of the last timed pass (outside the clock): size-independent invariants of the outputs
    check = {}
    if not (args.presplit and world > 1):
        out = eng.outputs_device()
        touched = ((o.qe.long() - 1) // p.reso - o.qs.long() // p.reso + 1).clamp(min=0)
        fo, fb, fe = out["frag_offset"], out["frag_begin"], out["frag_end"]
        same = out["frag_read"][1:] == out["frag_read"][:-1]
        check = {"sum_cov_equals_windows_touched": int(out["cov"].sum(dtype=torch.int64)) == int(touched.sum()) == s.total_coverage,
                 "fragments_tile_reads": bool((fb[fo[:-1]] == 0).all()) and bool((fe[fo[1:] - 1] == o.read_len).all())
                 and bool(((fe[:-1] - fb[1:])[same] == p.overlap_length).all()),
                 "windows": s.n_bins == int(((o.read_len.long() + p.reso - 1) // p.reso).sum())}
        del touched, out
        if not all(check.values()):
            raise SystemExit(f"bench.py: self-check failed: {check}")

    # ---- the pass as the CLI and the host pipelines run it: the pileup kernel writes the transfer encoding of cov[] (one
    # byte per window, two from -e 40 on, + the windows at or above the limit) instead of int32.  Same job, a third of the
    # HBM bytes; reported beside the int32 line, checked against it here (outside the clock).
    packed = None
    if n_gpus == 1 and not args.presplit and not args.no_packed_leg and args.cov_width == 4:
        w = 2 if p.est_cov >= 40 else 1
        e3 = engine.Engine(p_run, device=local)
        e3.set_tuning(args.tile_bins, args.force_bucket, args.variant)
        e3.set_output_width(w)
        e3.use_torch_stream()
        kt, pt, wall = [], [], []
        for it in range(6):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e3.run_device(*cols)
            s3 = e3.finish()
            torch.cuda.synchronize()
            if it:
                wall.append(time.perf_counter() - t1)
                a, b = e3.timing(); kt.append(a); pt.append(b)
        pk = e3.packed_device()
        assert pk is not None and pk["width"] == w
        ref_cov = eng.outputs_device()["cov"]
        lim = 255 if w == 1 else 65535
        codes = pk["cov8"] if w == 1 else pk["cov8"].to(torch.int32) & 0xFFFF
        ok = bool((codes == ref_cov.clamp(max=lim)).all())
        big = (ref_cov >= lim).nonzero().flatten()
        order = pk["exc_index"].argsort()
        ok = ok and bool(torch.equal(pk["exc_index"][order], big)) and bool(torch.equal(pk["exc_value"][order], ref_cov[big]))
        ok = ok and (s3.n_fragments, s3.n_repeats, s3.total_coverage, s3.total_repeat_length) == (s.n_fragments, s.n_repeats, s.total_coverage, s.total_repeat_length)
        if not ok:
            raise SystemExit("bench.py: the packed-output pass differs from the int32 pass")
        bytes_p = 12 * s3.n_intervals + w * s3.n_bins + 12 * int(pk["exc_index"].numel()) + 4 * s3.n_reads + 8 * s3.n_repeats
        k_s, p_s, w_s = sum(kt) / len(kt), sum(pt) / len(pt), sum(wall) / len(wall)
        packed = {"cov_width": w, "value": my_rec / w_s, "unit": "PAF records/s", "ms_per_step": w_s * 1e3, "kernel_ms": k_s * 1e3,
                  "pass_device_ms": p_s * 1e3, "bytes_algorithmic": bytes_p, "kernel_frac": bytes_p / k_s / 1e9 / HBM_PEAK_GBS,
                  "pass_frac": bytes_p / p_s / 1e9 / HBM_PEAK_GBS, "n_exceptions": int(pk["exc_index"].numel()),
                  "equals_int32_pass": ok}
        del ref_cov, codes, big, order, pk
        e3.close()

    inspect_ms = None
    if n_gpus == 1 and not args.presplit and not args.no_detect_leg:
        # the same pass in its round-1 form: inspect_kernel looks at every record before anything else starts
        os.environ["RAFT_ALWAYS_INSPECT"] = "1"
        try:
            e2 = engine.Engine(p_run, device=local)
            e2.set_tuning(args.tile_bins, args.force_bucket, args.variant)
            e2.use_torch_stream()
            tt = []
            for it in range(4):
                e2.run_device(*cols)
                s2 = e2.finish()
                if it:
                    tt.append(e2.timing()[1])
            assert (s2.n_fragments, s2.n_repeats, s2.total_coverage, s2.symmetric) == (s.n_fragments, s.n_repeats, s.total_coverage, 1)
            inspect_ms = sum(tt) / len(tt) * 1e3
            e2.close()
        finally:
            del os.environ["RAFT_ALWAYS_INSPECT"]

    if rank == 0:
        per_step = elapsed / args.steps
        # dominant kernel: pileup + prefix scan + coverage store + run scan (pileup.hpp).
        # algorithmic bytes per launch (this rank): 12 B per interval read once, 4 B per window written once,
        # 4 B per read (length) and 8 B per repeat emitted
        bytes_alg = 12 * s.n_intervals + args.cov_width * s.n_bins + 4 * s.n_reads + 8 * s.n_repeats
        pile = sum(pile_t) / len(pile_t)
        pass_dev = sum(pass_t) / len(pass_t)
        achieved = bytes_alg / pile / 1e9
        line = {
            "metric": "PAF overlap records/s + fragments/s, 32x human all-vs-all; HBM GB/s vs peak",
            "value": tot_rec / per_step, "unit": "PAF records/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "fragments_per_s": tot_frag / per_step,
            "config": {"workload": workload_text, "workload_name": args.workload,
                       "reads_per_gpu": s.n_reads, "records_per_gpu": my_rec, "records_total": tot_rec,
                       "windows_total": tot_bins, "intervals_total": tot_iv, "fragments_total": tot_frag,
                       "repeats_rank0": s.n_repeats, "mean_read_len": gen_kw["mean_len"], "coverage": gen_kw["coverage"],
                       "interval_path": "sorted-segments" if s.interval_path == 0 else "counting-sort",
                       "segments": s.n_segments,
                       "sharding": (f"reads x{n_gpus}, records pre-split, one all-to-all-v per step" if args.presplit and world > 1
                                    else f"reads x{n_gpus}, no data-path collective") + (", ranks share GPUs (gloo check run)" if shared else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": ("pileup_fast_kernel (+ pileup_kernel for the tiles it leaves)" if args.variant != 1 else "pileup_kernel")
                                   + ("" if args.variant < 0 else f" variant {args.variant}"),
                         "kernel_ms": pile * 1e3, "bytes_algorithmic": bytes_alg,
                         "pass_device_ms": pass_dev * 1e3, "pass_achieved": bytes_alg / pass_dev / 1e9,
                         "pass_frac": bytes_alg / pass_dev / 1e9 / HBM_PEAK_GBS, "kernel_source_hash": kernel_source_hash(),
                         "symmetric_mode": "handed over by the tokeniser (as the CLI does)" if args.handover else "detected by the engine (mirror of record 0 found beside the pileup)",
                         "pass_device_ms_inspect_first": inspect_ms},
            "self_check": check,
        }
        if packed is not None:
            line["packed_output"] = packed
        if args.cov_width != 4:
            line["config"]["cov_width"] = args.cov_width
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
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
        if n_gpus == 1 and not args.presplit:
            if not args.no_e2e:
                try:
                    line["e2e"] = e2e_leg(args, torch, engine, o, p)
                except engine.RaftError as ex:
                    # e.g. the ultralong workload: tandem arrays at 6 copies put half of all windows at or above 255, where
                    # the byte-per-window transfer encoding has to list them one by one -- more than the leg's buffers hold
                    line["e2e"] = {"error": str(ex), "note": "transfer encoding of cov[] (uint8 + exceptions) not suited to this "
                                   "coverage: the caller has to provide room for one exception per window at or above 255"}
            if not args.no_cpu_baseline:
                cb, _ = cpu_baseline(args, o, p)
                line["cpu_baseline"] = cb
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
