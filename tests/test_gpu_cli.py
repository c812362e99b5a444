"""GPU: the `raft` command line (raft_amd/bin/raft -> libraft_hip.so) against the reference's own outputs:
the four files byte for byte, stdout line for line (minus the timing / CMD lines), exit codes."""
import json
import os
import shutil
import subprocess

import pytest
from raft_testlib import GOLDEN, ROOT, md5, write_fasta, write_paf
from test_oracle_golden import MAN, load_case

pytestmark = pytest.mark.gpu
RAFT = os.path.join(ROOT, "raft_amd", "bin", "raft")


def strip_timing(out: str) -> str:
    return "\n".join(l for l in out.split("\n")
                     if not l.startswith("INFO, main(), program completed after") and not l.startswith("INFO, main(), CMD:"))


def run(cwd, args, env=None):
    r = subprocess.run([RAFT] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                       env=dict(os.environ, **env) if env else None)
    return r.returncode, r.stdout.decode()


# the C++ host on several contexts / chunked: RAFT_DEVICES names the GPUs that share the job (the box has one, so it is
# named twice or three times: same code path, two or three contexts), RAFT_CHUNKS forces the chunked pipeline on small inputs
MULTI = [None, {"RAFT_CHUNKS": "3"}, {"RAFT_DEVICES": "0,0", "RAFT_CHUNKS": "5"}, {"RAFT_DEVICES": "0,0,0", "RAFT_CHUNKS": "7"}]


@pytest.mark.parametrize("name", sorted(MAN["micro"]))
def test_cli_micro_cases(tmp_path, name):
    d = os.path.join(GOLDEN, "micro", name)
    meta = MAN["micro"][name]
    shutil.copy(os.path.join(d, "reads.fa"), tmp_path)
    shutil.copy(os.path.join(d, "overlaps.paf"), tmp_path)
    rc, out = run(tmp_path, meta["args"] + ["reads.fa", "overlaps.paf"])
    assert rc == 0, out
    assert strip_timing(out) == open(os.path.join(d, "expect.stdout")).read()
    assert "INFO, main(), program completed after" in out and "INFO, main(), CMD: " + RAFT in out
    produced = sorted(f for f in os.listdir(tmp_path) if f not in ("reads.fa", "overlaps.paf"))
    assert produced == meta["outputs"]
    for f in produced:
        assert open(tmp_path / f, "rb").read() == open(os.path.join(d, "expect." + f), "rb").read(), (name, f)


@pytest.mark.parametrize("env", MULTI, ids=lambda e: "one" if e is None else "-".join(f"{k[5:]}{v}" for k, v in e.items()))
@pytest.mark.parametrize("name", sorted(MAN["synthetic"]))
def test_cli_synthetic_cases(tmp_path, name, env):
    p, cols, exp, meta = load_case(name)
    names = [f"r{i}" for i in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    write_paf(tmp_path / "overlaps.paf", names, *cols)
    rc, out = run(tmp_path, meta["args"] + ["reads.fa", "overlaps.paf"], env)
    assert rc == 0, out
    assert strip_timing(out) == meta["stdout"]
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, (name, f)


@pytest.mark.parametrize("env", [None, {"RAFT_DEVICES": "0,0", "RAFT_CHUNKS": "6"}], ids=["one", "two-contexts"])
def test_cli_config1_gz(tmp_path, env):
    """BASELINE configs[0] (README.md:12-33 `raft -e 42 -o fragmented reads.fa.gz overlaps.paf`) on the SURVEY §8(d)
    stand-in (2 Mbp, 42x), both inputs gzip-compressed: four files md5-identical to the reference's, stdout equal."""
    from raft_testlib import load_config1, write_config1_inputs
    p, cols, exp, meta = load_config1()
    write_config1_inputs(str(tmp_path), cols, meta)
    rc, out = run(tmp_path, meta["args"] + meta["inputs"], env)
    assert rc == 0, out
    assert strip_timing(out) == meta["stdout"]
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("fragmented." + f), "rb").read()) == digest, f


@pytest.mark.parametrize("i", range(0, 320, 13))
def test_cli_ref_fuzz_subset(tmp_path, i):
    """Every 13th case of tests/golden/ref_fuzz.npz through the CLI: the four files md5-identical to the reference's."""
    from raft_testlib import ref_fuzz_case
    p, cols, exp = ref_fuzz_case(i)
    names = [f"r{k}" for k in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    write_paf(tmp_path / "overlaps.paf", names, *cols)
    rc, out = run(tmp_path, p.cli_args() + ["-o", "out", "reads.fa", "overlaps.paf"])
    assert rc == 0, out
    assert "INFO, Symmetric overlaps %d \n" % exp["symmetric"] in out
    for f, digest in exp["md5"].items():
        assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, (i, f)


def test_cli_usage_and_input_errors(tmp_path):
    rc, out = run(tmp_path, [])
    assert rc == 1 and out.startswith("Usage: raft [options] <input-reads.fa> <in.paf>\n")
    rc, out = run(tmp_path, ["-e", "30", "-i", "5", "a.fa", "b.paf"])       # -i is in the getopt string but has no case
    assert rc == 1 and "Usage: raft" in out
    rc, out = run(tmp_path, ["a.fa", "b.paf"])                               # est_cov not set
    assert rc == 1 and out.startswith("ERROR, main(), estimated coverage must be set properly\nUsage:")
    rc, out = run(tmp_path, ["-e", "30", "-o", "pre", "a.fa", "b.paf"])
    assert rc == 1 and "ERROR, break_long_reads(), a.fa input file either does not exist or is empty" in out
    assert os.path.exists(tmp_path / "pre.reads.fasta") and os.path.getsize(tmp_path / "pre.reads.fasta") == 0
    (tmp_path / "a.fa").write_text(">x\nACGT\n")
    (tmp_path / "b.paf").write_text("x\t4\t0\t4\t+\tnope\t4\t0\t4\t1\t1\t1\n")
    rc, out = run(tmp_path, ["-e", "30", "a.fa", "b.paf"])
    assert rc == 1 and "read nope of the overlaps file is not in the reads file" in out
    (tmp_path / "b.paf").write_text("x\t4\t0\t400\t+\tx\t4\t0\t4\t1\t1\t1\n")  # reaches past the last window: defined error
    rc, out = run(tmp_path, ["-e", "30", "a.fa", "b.paf"])
    assert rc == 1 and "ERROR, raft_hip, PAF coordinate" in out
    rc, out = run(tmp_path, ["-e", "30", "a.fa", "b.paf"], {"RAFT_DEVICES": "0,7"})   # a device that does not exist
    assert rc == 1 and "ERROR, raft_hip_create(), device 7" in out


BOUND = os.path.join(ROOT, "oracle", "_ref", "raft_bound")


@pytest.mark.skipif(not os.path.exists(BOUND), reason="oracle/_ref/raft_bound not built (needs /root/reference: build container)")
@pytest.mark.parametrize("name", ["g1", "g2", "g3", "g4", "s300_default", "s300_nonsym_shuffled", "edge_reads"])
def test_reference_side_binding(tmp_path, name):
    """INTEGRATION.md §B compiled (oracle/ref_binding.cpp): the reference's OWN loaders, name table and types, with the
    three hot calls of break_long_reads() replaced by the C ABI -- same four files as the unmodified reference."""
    if name in MAN["micro"]:
        d = os.path.join(GOLDEN, "micro", name)
        meta = MAN["micro"][name]
        shutil.copy(os.path.join(d, "reads.fa"), tmp_path)
        shutil.copy(os.path.join(d, "overlaps.paf"), tmp_path)
        args, want_stdout = meta["args"], open(os.path.join(d, "expect.stdout")).read()
        expect = {f: open(os.path.join(d, "expect." + f), "rb").read() for f in meta["outputs"]}
    else:
        p, cols, exp, meta = load_case(name)
        names = [f"r{i}" for i in range(len(cols[0]))]
        write_fasta(tmp_path / "reads.fa", names, cols[0])
        write_paf(tmp_path / "overlaps.paf", names, *cols)
        args, want_stdout, expect = meta["args"], meta["stdout"], None
    r = subprocess.run([BOUND] + args + ["reads.fa", "overlaps.paf"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-500:]
    assert strip_timing(r.stdout.decode()) == want_stdout
    if expect is not None:
        for f, want in expect.items():
            assert open(tmp_path / f, "rb").read() == want, (name, f)
    else:
        for f, digest in meta["md5"].items():
            assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, (name, f)


@pytest.mark.parametrize("name", ["s300_nonsym_shuffled", "s300_sym_shuffled", "s300_default"])
def test_cli_spreads_every_input_shape_over_the_contexts(tmp_path, name):
    """RAFT_DEVICES names two contexts: a hifiasm-shaped PAF is cut into chunks of its sorted runs (grouped input), any
    other stream -- shuffled, non-symmetric -- is bucketed by the host's threads and routed (engine.hip run_routed).  Both
    contexts take part (the stage clock on stderr says so) and the four files are the reference's."""
    p, cols, exp, meta = load_case(name)
    names = [f"r{i}" for i in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    write_paf(tmp_path / "overlaps.paf", names, *cols)
    r = subprocess.run([RAFT] + meta["args"] + ["reads.fa", "overlaps.paf"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=dict(os.environ, RAFT_DEVICES="0,0", RAFT_CHUNKS="4", RAFT_TIMING="1"))
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert strip_timing(r.stdout.decode()) == meta["stdout"]
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, (name, f)
    line = [l for l in r.stderr.decode().splitlines() if l.startswith("TIMING devices_used")]
    assert line and line[0].split()[2] == "2", r.stderr.decode()
    assert line[0].split()[4] == "columns"           # (round 4: the engine derives what a hifiasm-shaped stream allows by itself)
    assert ("derived by the engine" in line[0]) == (name != "s300_nonsym_shuffled")


@pytest.mark.parametrize("prepare", [False, True])
def test_cli_many_concatenated_files(tmp_path, prepare):
    """Six PAFs concatenated (each grouped by query): more sorted runs than the pileup kernels take.  The CLI still hands the
    stream over in grouped form -- the engine merges the runs on the device -- and the files equal those of the same records
    in one sorted file, which the golden case pins to the reference."""
    import numpy as np
    p, cols, exp, meta = load_case("s300_default")
    names = [f"r{i}" for i in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    n = len(cols[1])
    # the same multiset of records as six runs: record i goes to file i % 6, every file keeps the stream's order
    order = np.concatenate([np.arange(k, n, 6) for k in range(6)])
    # (record 0 stays first: it decides the symmetric flag, chop.hpp:171-184)
    write_paf(tmp_path / "overlaps.paf", names, cols[0], *[c[order] for c in cols[1:]])
    r = subprocess.run([RAFT] + meta["args"] + ["reads.fa", "overlaps.paf"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=dict(os.environ, RAFT_TIMING="1", **({"RAFT_CLI_PREPARE": "1"} if prepare else {})))
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert ("input windows" if prepare else "input columns") in r.stderr.decode()
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, f


@pytest.mark.parametrize("knob,form", [("RAFT_NO_WINDOWS", "grouped"), ("RAFT_NO_GROUPED", "columns"), (None, "windows"), ("RAFT_NO_DELTA4", "windows"),
                                       ("default", "columns (offsets and window records derived by the engine)")])
@pytest.mark.parametrize("name", ["s300_default", "s60_ultralong"])
def test_cli_input_forms(tmp_path, name, knob, form):
    """The three forms the CLI hands a symmetric hifiasm-shaped PAF over in -- window records (default), coordinate columns
    with the per-read offsets, the plain columns -- give the reference's files.  s60_ultralong runs at -r 10 with reads of
    more than 65,535 windows: window records cannot say that, and the CLI stays with the coordinate columns by itself."""
    p, cols, exp, meta = load_case(name)
    names = [f"r{i}" for i in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    write_paf(tmp_path / "overlaps.paf", names, *cols)
    env = dict(os.environ, RAFT_TIMING="1")
    if knob != "default":
        env["RAFT_CLI_PREPARE"] = "1"                 # (the forms prepared by the host library in front of the engine; the default hands over the plain columns)
        if knob:
            env[knob] = "1"
    r = subprocess.run([RAFT] + meta["args"] + ["reads.fa", "overlaps.paf"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert strip_timing(r.stdout.decode()) == meta["stdout"]
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, (name, f)
    if name == "s60_ultralong" and form == "windows":
        form = "grouped"
    assert f"input {form}" in r.stderr.decode(), r.stderr.decode()
    # grouped input brings the coverage back as four-bit steps unless told otherwise; the plain columns keep the byte encodings
    enc = "delta4" if (form != "columns" and knob != "RAFT_NO_DELTA4") else ("uint16" if p.est_cov >= 40 else "uint8")
    assert f"coverage_encoding {enc}" in r.stderr.decode(), r.stderr.decode()


@pytest.mark.parametrize("ranks", [2, 3])
@pytest.mark.parametrize("name", ["s300_nonsym_shuffled", "s300_default", "g3"])
def test_cli_presplit_ranks(tmp_path, name, ranks):
    """RAFT_RANKS=N: the pre-split job (BASELINE configs[3]) behind the reference's command line -- N ranks as
    contexts of the one GPU, each holding a contiguous slice of the record stream, ONE exchange step (raft_hip_run_presplit_local).
    The four files and stdout are the single-process run's = the reference's, byte for byte."""
    if name in MAN["micro"]:
        d = os.path.join(GOLDEN, "micro", name)
        meta = MAN["micro"][name]
        shutil.copy(os.path.join(d, "reads.fa"), tmp_path)
        shutil.copy(os.path.join(d, "overlaps.paf"), tmp_path)
        want_out = open(os.path.join(d, "expect.stdout")).read()
        want_files = {f: md5(open(os.path.join(d, "expect." + f), "rb").read()) for f in meta["outputs"]}
    else:
        p, cols, exp, meta = load_case(name)
        names = [f"r{i}" for i in range(len(cols[0]))]
        write_fasta(tmp_path / "reads.fa", names, cols[0])
        write_paf(tmp_path / "overlaps.paf", names, *cols)
        want_out = meta["stdout"]
        want_files = {"out." + f: digest for f, digest in meta["md5"].items()}
    r = subprocess.run([RAFT] + meta["args"] + ["reads.fa", "overlaps.paf"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=300, env=dict(os.environ, RAFT_RANKS=str(ranks), RAFT_TIMING="1"))
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert strip_timing(r.stdout.decode()) == want_out
    for f, digest in want_files.items():
        assert md5(open(tmp_path / f, "rb").read()) == digest, (name, f)
    line = [l for l in r.stderr.decode().splitlines() if l.startswith("TIMING devices_used")]
    assert line and line[0].split()[2] == str(ranks) and "pre-split" in line[0], r.stderr.decode()
