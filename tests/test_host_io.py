"""CPU: the host text layer (libraft_host.so) -- tokenisation equals the reference's, and the writers are byte-exact.

Pipeline under test: FASTA/PAF text -> C++ readers -> [oracle arrays in between, test-only] -> C++ writers ->
bytes compared with what the compiled reference wrote for the same inputs (tests/golden/).
"""
import gzip
import json
import os
import re
import shutil

import numpy as np
import pytest
from raft_testlib import GOLDEN, ROOT, RaftParams, md5, oracle_run, write_fasta, write_paf

from raft_amd import hostio
from test_oracle_golden import MAN, _micro_columns, _params_from_args, load_case


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raft_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(raft_host_[a-z_]+)\s*\(", text)))


def test_host_library_exports_every_symbol():
    lib = hostio.load_library()
    assert declared_symbols() == sorted(hostio.EXPORTS)
    for n in declared_symbols():
        assert hasattr(lib, n), n


def run_text_case(tmp_path, fa, paf, p, prefix):
    reads = hostio.Reads(str(fa))
    cols = hostio.load_paf(str(paf), reads)
    res = oracle_run(p, reads.lengths, *cols)
    hostio.write_outputs(str(tmp_path / prefix), reads, p.reso, res)
    return reads, cols, res


@pytest.mark.parametrize("name", sorted(MAN["micro"]))
@pytest.mark.parametrize("gz", [False, True])
def test_micro_cases_byte_exact(tmp_path, name, gz):
    d = os.path.join(GOLDEN, "micro", name)
    meta = MAN["micro"][name]
    fa, paf = os.path.join(d, "reads.fa"), os.path.join(d, "overlaps.paf")
    if gz:
        for src, dst in ((fa, tmp_path / "reads.fa.gz"), (paf, tmp_path / "overlaps.paf.gz")):
            with open(src, "rb") as i, gzip.open(dst, "wb") as o:
                shutil.copyfileobj(i, o)
        fa, paf = tmp_path / "reads.fa.gz", tmp_path / "overlaps.paf.gz"
    p = _params_from_args(meta["args"])
    prefix = [f for f in meta["outputs"] if f.endswith(".coverage.txt")][0][: -len(".coverage.txt")]
    reads, cols, res = run_text_case(tmp_path, fa, paf, p, prefix)
    # tokenisation agrees with an independent python tokeniser
    names, lens, pycols = _micro_columns(d)
    assert [reads.name(i) for i in range(reads.n)] == names and reads.lengths.tolist() == lens.tolist()
    for a, b in zip(cols, pycols):
        assert np.array_equal(a, b)
    for f in meta["outputs"]:
        got = open(tmp_path / f, "rb").read()
        want = open(os.path.join(d, "expect." + f), "rb").read()
        assert got == want, (name, f)


@pytest.mark.parametrize("name", sorted(MAN["synthetic"]))
def test_synthetic_cases_md5(tmp_path, name):
    p, cols, exp, meta = load_case(name)
    names = [f"r{i}" for i in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    write_paf(tmp_path / "overlaps.paf", names, *cols)
    reads, got_cols, res = run_text_case(tmp_path, tmp_path / "reads.fa", tmp_path / "overlaps.paf", p, "out")
    for a, b in zip(got_cols, cols[1:]):
        assert np.array_equal(a, b)
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("out." + f), "rb").read()) == digest, (name, f)


def test_fastq_and_odd_fasta(tmp_path):
    fq = tmp_path / "r.fq"
    fq.write_text("@q1 some comment\nACGTAC\nGT\n+q1\nIIIIII\nII\n@q2\nAAAA\n+\n!!!!\n>f3\tx\nCC\r\n\r\nGG\r\n")
    r = hostio.Reads(str(fq))
    assert [r.name(i) for i in range(r.n)] == ["q1", "q2", "f3"]
    assert r.lengths.tolist() == [8, 4, 4]
    assert r.bases(0) == b"ACGTACGT" and r.bases(2) == b"CCGG" and r.real == 1
    bad = tmp_path / "bad.fq"
    # quality lines are consumed until they are as long as the sequence, whatever they contain: "II" + "@b" closes
    # record a, then the reader resynchronises on the next '>' / '@' and finds none (verified with the reference)
    bad.write_text("@a\nACGT\n+\nII\n@b\nAC\n+\nII\n")
    rb = hostio.Reads(str(bad))
    assert rb.n == 1 and rb.name(0) == "a" and rb.bases(0) == b"ACGT"
    trunc = tmp_path / "trunc.fq"
    trunc.write_text("@a\nACGT\n+\nIIII\n@b\nACGT\n+\nII\n")   # qualities shorter than the sequence at EOF: record dropped
    assert hostio.Reads(str(trunc)).n == 1
    sim = tmp_path / "sim.fa"
    sim.write_text(">read=7,reverse,position=100-150,length=50,chrZ\n" + "A" * 50 + "\n>plain\nCC\n")
    s = hostio.Reads(str(sim))
    assert s.real == 0 and s.n == 2


def test_paf_number_semantics_and_errors(tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text(">a\nAAAAAAAAAA\n>b\nCCCCCCCCCC\n")
    reads = hostio.Reads(str(fa))
    paf = tmp_path / "o.paf"
    lines = ["a\t10\t 3\t7x\t+\tb\t10\t\t9\t1\t1\t1\n",          # ' 3' -> 3, '7x' -> 7, '' -> 0
             "a\t10\t4294967295\t5\t+\tb\t10\t1\t2\t1\n",        # 2^32-1 -> uint32 -> int -1
             "b\t10\t1\t2\t+\ta\t10\t1\t2\t1"]                   # last line without newline
    paf.write_text("".join(lines))
    cols = hostio.load_paf(str(paf), reads)
    assert cols[0].tolist() == [0, 0, 1] and cols[1].tolist() == [3, -1, 1] and cols[2].tolist() == [7, 5, 2]
    assert cols[4].tolist() == [0, 1, 1] and cols[5].tolist() == [9, 2, 2]
    paf.write_text("a\t10\t1\t2\t+\tzzz\t10\t1\t2\t1\n")
    with pytest.raises(hostio.HostError) as e:
        hostio.load_paf(str(paf), reads)
    assert e.value.code == hostio.ERR_UNKNOWN_NAME and e.value.what == "zzz"
    dup = tmp_path / "dup.fa"
    dup.write_text(">a\nAA\n>a\nCC\n")
    with pytest.raises(hostio.HostError) as e:
        hostio.Reads(str(dup))
    assert e.value.code == hostio.ERR_DUP_NAME
    with pytest.raises(hostio.HostError) as e:
        hostio.Reads(str(tmp_path / "missing.fa"))
    assert e.value.code == hostio.ERR_OPEN
