#!/usr/bin/env python3
"""bench.py -- RAFT hot path (PAF overlaps -> coverage -> repeat mask -> fragments) on MI355X.

One "step" = one full pass of the engine (raft_hip_run_device + raft_hip_finish) over one
synthetic all-vs-all overlap set that is already resident in HBM when the clock starts:
record inspection, per-tile interval ranges (or counting-sort bucketing), the pileup /
prefix-sum / run-scan kernel, repeat ordering, cut points, fragment table and the stdout
statistics.  Outputs stay in HBM (DESIGN.md gives the PCIe-inclusive rate separately).

Workload (config.workload): BASELINE.json configs[2] restated synthetically (SURVEY.md §8d,
config 3): HG002-like 32x set, 3.3 M reads of 30 kb mean length, ~1e8 symmetric PAF records
written as a cis file followed by a trans file, each grouped by ascending query id.  With
--gpus N every rank owns an independent shard of that size (reads and their overlaps shard
embarrassingly; no data-path collective) -- weak scaling; the only collective is the
all-gather of per-rank fragment totals that turns local fragment ids into global read_num.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def cpu_baseline(args, torch, make_overlaps, RaftParams):
    """Times the CPU checkers on a bounded sample of the same workload (same generator, fewer reads).

    kind "reference": the unmodified reference's own code (create_pileup's bucket fill restated in
    oracle/ref_harness.cpp + repeat_annotate/profileCoverage compiled from /root/reference into
    oracle/_ref/libraft_ref.so in the build container) -- single thread, as the reference ships.
    Falls back to kind "port" (oracle/raft_oracle.c) when the prebuilt reference harness is absent.
    """
    import raft_testlib as tl
    n = args.cpu_sample_reads
    o = make_overlaps(n, mean_len=args.mean_len, coverage=args.coverage, seed=args.seed + 1000, device="cuda:0")
    cols = [c.cpu().numpy() for c in (o.read_len,) + o.columns()]
    p = RaftParams(est_cov=int(args.coverage))
    t0 = time.perf_counter()
    res = tl.oracle_run(p, *cols)
    t_port = time.perf_counter() - t0
    out = {"cores": 1, "unit": "PAF records/s",
           "sample": f"{n} reads / {o.n_rec} records of the same generator (seed {args.seed + 1000}), single thread",
           "port_records_per_s": o.n_rec / t_port, "port_seconds": t_port,
           "port_fragments_per_s": len(res["frag_read"]) / t_port}
    if tl.have_ref_lib():
        r = tl.ref_lib_run(p, *cols, want_cov=False)
        t_ref = r["seconds_bucket"] + r["seconds_annotate"]
        out.update(kind="reference", value=o.n_rec / t_ref, reference_seconds=t_ref,
                   note="reference = bucket fill (chop.hpp:155-184 restated) + repeat_annotate() of the unmodified "
                        "reference (profileCoverage + run scan, text streams disabled); break_reads' integer half is "
                        "not separately callable in the reference and is not in this time")
    else:
        out.update(kind="port", value=o.n_rec / t_port)
    return out, (o, p, res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=3_300_000, help="reads per GPU")
    ap.add_argument("--mean-len", type=float, default=30000.0)
    ap.add_argument("--coverage", type=float, default=32.0)
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--cpu-sample-reads", type=int, default=150_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-bins", type=int, default=0)
    ap.add_argument("--force-bucket", action="store_true")
    ap.add_argument("--variant", type=int, default=-1, help="pileup kernel variant (engine.hip kVariants), -1 = default")
    args = ap.parse_args()

    import torch

    from raft_amd import engine
    from raft_amd.params import RaftParams
    from raft_amd.synth import make_overlaps

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    dist = None
    coll_dev = None
    if world > 1:
        import torch.distributed as dist
        if n_dev >= world:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))     # RCCL over xGMI
            coll_dev = f"cuda:{local}"
        else:
            # fewer GPUs than ranks (a 1-GPU box): ranks share devices and the tiny collectives go through gloo --
            # only good for checking the multi-rank code path, not a scaling number
            local = local % max(n_dev, 1)
            dist.init_process_group("gloo")
            coll_dev = "cpu"
    n_gpus = max(world, 1)
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)

    # ---- synthetic shard, generated on the device (resident in HBM before the clock starts)
    p = RaftParams(est_cov=int(args.coverage))
    o = make_overlaps(args.reads, mean_len=args.mean_len, coverage=args.coverage, seed=args.seed + rank, device=dev)
    cols = (o.read_len,) + o.columns()
    torch.cuda.synchronize()

    eng = engine.Engine(p, device=local)
    eng.set_tuning(args.tile_bins, args.force_bucket, args.variant)
    eng.use_torch_stream()
    from raft_amd import dist as rdist

    def step():
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
        cnt = torch.tensor([o.n_rec, s.n_fragments, s.n_bins, s.n_intervals], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_rec, tot_frag, tot_bins, tot_iv = (int(x) for x in cnt.tolist())
    else:
        tot_rec, tot_frag, tot_bins, tot_iv = o.n_rec, s.n_fragments, s.n_bins, s.n_intervals

    if rank == 0:
        per_step = elapsed / args.steps
        # dominant kernel: pileup + prefix scan + coverage store + run scan (pileup.hpp).
        # algorithmic bytes per launch (this rank): 12 B per interval read once, 4 B per window written once,
        # 4 B per read (length) and 8 B per repeat emitted
        bytes_alg = 12 * s.n_intervals + 4 * s.n_bins + 4 * s.n_reads + 8 * s.n_repeats
        pile = sum(pile_t) / len(pile_t)
        achieved = bytes_alg / pile / 1e9
        line = {
            "metric": "PAF overlap records/s + fragments/s, 32x human all-vs-all; HBM GB/s vs peak",
            "value": tot_rec / per_step, "unit": "PAF records/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "fragments_per_s": tot_frag / per_step,
            "config": {"workload": "HG002-like 32x all-vs-all PAF restated synthetically (BASELINE configs[2]): cis+trans "
                                   "files grouped by query id, symmetric, raft -e 32 defaults (-r 50 -p 10000 -l 20000 -f 1000 -v 500)",
                       "reads_per_gpu": o.n_reads, "records_per_gpu": o.n_rec, "records_total": tot_rec,
                       "windows_total": tot_bins, "intervals_total": tot_iv, "fragments_total": tot_frag,
                       "repeats_rank0": s.n_repeats, "mean_read_len": args.mean_len, "coverage": args.coverage,
                       "interval_path": "sorted-segments" if s.interval_path == 0 else "counting-sort",
                       "segments": s.n_segments, "sharding": f"reads x{n_gpus}, no data-path collective" + (", ranks share GPUs (gloo check run)" if world > 1 and n_dev < world else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": ("pileup_fast_kernel (+ pileup_kernel for the tiles it leaves)" if args.variant != 1 else "pileup_kernel")
                                   + ("" if args.variant < 0 else f" variant {args.variant}"),
                         "kernel_ms": pile * 1e3, "bytes_algorithmic": bytes_alg,
                         "pass_device_ms": sum(pass_t) / len(pass_t) * 1e3},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            cb, _ = cpu_baseline(args, torch, make_overlaps, RaftParams)
            line["cpu_baseline"] = cb
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                if tj.get("records_per_gpu") == o.n_rec:
                    line["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = tj.get("source")
            except Exception:
                pass
        print(json.dumps(line))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
