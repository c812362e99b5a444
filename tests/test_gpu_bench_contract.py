"""GPU: bench.py keeps its one-line JSON contract (fields the driver and the judge read)."""
import json
import os
import subprocess
import sys

import pytest
from raft_testlib import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_has_every_contract_field():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "60000", "--steps", "2", "--warmup", "1",
                        "--cpu-sample-reads", "3000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]   # (one GPU: nothing to shard)
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # round 2: whole-pass roofline, end-to-end leg, self-check, CPU description
    assert 0 < rf["pass_frac"] <= rf["frac"] and rf["pass_device_ms"] >= rf["kernel_ms"] and len(rf["kernel_source_hash"]) == 16
    e = d["e2e"]
    for k in ("records_per_s", "fragments_per_s", "seconds", "first_pass_s", "h2d_bytes", "d2h_bytes", "host_memory"):
        assert k in e, k
    assert e["decoded_coverage_equals_device"] is True and e["chunked_equals_one_piece"] is True and e["records_per_s"] > 0 and e["d2h_bytes"] < e["h2d_bytes"] * 3
    assert all(d["self_check"].values()) and len(d["self_check"]) == 4
    # round 4: the headline is the self-contained six-column pass into a detecting context again (the grouped form -- part of
    # create_pileup's bucketing handed over by the caller -- is a leg beside it), and the pass writes the cut points itself
    assert "six plain columns" in d["config"]["input"] and "detecting" in d["config"]["input"] and "inside the step" in d["config"]["cut_points"]
    assert d["config"]["cut_points_total_rank0"] >= d["config"]["reads_per_gpu"] and rf["pass_includes_cut_points"] is True
    assert 0 < rf["pass_device_ms_without_cuts"] < rf["pass_device_ms"] * 1.2
    # the product path (no query column, a byte per window) sits in `roofline`; the grouped form and the six-column host pipeline are timed beside it
    for k in ("product_path_kernel_ms", "product_path_frac", "product_path_pass_frac"):
        assert rf[k] > 0, k
    assert d["packed_output"]["equals_int32_pass"] is True and "no query column" in d["packed_output"]["input"]
    assert "six_column" not in d and d["grouped"]["pass_device_ms"] > 0 and d["grouped"]["six_column_pass_device_ms_inspect_first"] > 0
    assert e["six_column_input"]["equals_grouped"] is True and e["six_column_input"]["h2d_bytes"] > e["h2d_bytes"]
    assert cb["cpu_model"] and cb["node_logical_cpus"] >= 1 and "same seed" in cb["sample"]
    # ... window records and the four-bit step encoding of the coverage: the CLI's forms, each checked against the int32 pass
    assert rf["product_path_input"] == "window records"
    assert d["window_records"]["equals_int32_pass"] is True and d["window_records_delta4"]["equals_int32_pass"] is True
    assert d["window_records_delta4"]["n_exceptions"] > 0 and d["window_records_delta4"]["cov_width"] == 8
    w = e["window_records"]
    assert w["equals_coordinate_columns"] is True and w["delta4"]["decoded_equals_byte_encoding"] is True
    assert w["delta4"]["d2h_bytes"] < e["byte_per_window_d2h_bytes"] and e["coordinate_columns"]["h2d_bytes"] > e["h2d_bytes"]
    # the object's headline starts at SURVEY.md §8(d)'s boundary: the grouped form and the window records are derived inside the clock
    fs = w["from_soa"]
    assert fs["equals_prepared_input"] is True and e["records_per_s"] == fs["records_per_s"] and e["seconds"] == fs["seconds"]
    assert fs["explicit_host_calls"]["seconds"] > 0 and "ONE call" in fs["boundary"]
    assert e["prepared_input"]["records_per_s"] == w["delta4"]["records_per_s"] and "SoA boundary" in e["mode"]


@pytest.mark.parametrize("extra", [[], ["--host-routed"], ["--weak"]])
def test_bench_gpus_flag_spawns_the_ranks_itself(extra):
    """`python bench.py --gpus 2` without a launcher: two ranks (sharing the box's one GPU, gloo) and n_gpus == 2 in the line.
    Round 6: the headline of a multi-GPU run IS BASELINE configs[3] -- the ONE set, reads owned in contiguous ranges, the records
    pre-split across the ranks and routed by one all-to-all-v inside every step, "scaling": "strong" -- and the other two forms
    (host-routed: same set, no exchange; weak: a set per rank) are legs of the same line; --host-routed / --weak swap the roles."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "30000", "--steps", "2",
                        "--warmup", "1"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["config"]["records_total"] > d["config"]["records_per_gpu"] > 0
    assert "cpu_baseline" not in d and "e2e" not in d
    head = "weak" if extra == ["--weak"] else ("host_routed" if extra else "presplit")
    assert set(d) >= {"presplit", "host_routed", "weak"} - {head} and head not in d
    for form in {"presplit", "host_routed"} - {head}:
        assert d[form]["scaling"] == "strong" and d[form]["ranks_totals_equal_single_gpu_pass"] is True and d[form]["value"] > 0
        assert d[form]["rank0_records"] < d[form]["records_total"]
    if head == "weak":
        assert d["scaling"] == "weak" and "independent shards" in d["config"]["sharding"]
        assert d["presplit"]["records_total"] == d["config"]["records_per_gpu"]
    else:                                                 # BASELINE configs[3]: ONE set, sharded
        assert d["scaling"] == "strong" and d["self_check"]["ranks_totals_equal_single_gpu_pass"] is True
        assert ("pre-split" in d["config"]["sharding"]) == (head == "presplit") and ("host-routed" in d["config"]["sharding"]) == (head == "host_routed")
        assert d["config"]["records_total"] == d[("host_routed" if head == "presplit" else "presplit")]["records_total"]
        w = d["weak"]
        assert w["scaling"] == "weak" and w["value"] > 0 and w["records_total"] > 1.5 * d["config"]["records_total"]


def test_bench_watchdog_ends_a_job_whose_rank_fails():
    """A rank that dies must not leave its peer waiting in a collective for ever: the parent ends the others and the job exits
    non-zero (here: rank 1 is told to exit before its first collective)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RAFT_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "20000", "--steps", "1", "--warmup", "0",
                        "--no-extra-legs"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.decode().splitlines() if l.startswith("{")]


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "1000"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600, env=env)
    assert r.returncode != 0 and b"WORLD_SIZE" in r.stderr
