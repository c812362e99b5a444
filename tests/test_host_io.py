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
    return sorted(set(re.findall(r"\b(raft_host_[a-z0-9_]+)\s*\(", text)))


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


def test_config1_gz_inputs_md5(tmp_path):
    """BASELINE configs[0] stand-in: gz FASTA + gz PAF through the host readers (zlib path), oracle in between,
    writers -> the md5 of the four files the reference wrote for the same gz inputs."""
    from raft_testlib import load_config1, write_config1_inputs
    p, cols, exp, meta = load_config1()
    write_config1_inputs(str(tmp_path), cols, meta)
    reads, got_cols, res = run_text_case(tmp_path, tmp_path / "reads.fa.gz", tmp_path / "overlaps.paf.gz", p, "fragmented")
    for a, b in zip(got_cols, cols[1:]):
        assert np.array_equal(a, b)
    for f, digest in meta["md5"].items():
        assert md5(open(tmp_path / ("fragmented." + f), "rb").read()) == digest, f
    for threads in (1, 5):                               # 25 MB of PAF text: the multi-chunk tokeniser finds the flag too
        hostio.set_threads(threads)
        assert hostio.load_paf(str(tmp_path / "overlaps.paf.gz"), reads, with_flag=True)[1] == meta["symmetric"] == 1
        # the two-step form the CLI uses (text fetched beside the reads, then parsed): the same columns and flag
        two, sym2 = hostio.load_paf(str(tmp_path / "overlaps.paf.gz"), reads, with_flag=True, two_steps=True)
        assert sym2 == 1 and all(np.array_equal(a, b) for a, b in zip(two, cols[1:]))
    hostio.set_threads(0)


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


# ---- worker threads: the parallel loaders/writers give the bytes of the sequential ones -------------------------

def _snapshot_reads(path):
    r = hostio.Reads(str(path))
    snap = ([r.name(i) for i in range(r.n)], r.lengths.tolist(), [r.bases(i) for i in range(r.n)], r.real)
    r.close()
    return snap


FASTA_SHAPES = {
    "wrapped": ">a c1 c2\nACGT\nAC\n>b\tx\nGG\n\nTT\n>c\n>d\nA\n",
    "no_final_newline": ">a\nACGT\n>b\nGGTT",
    "bare_header_at_eof": ">a\nACGT\n>",
    "name_at_eof": ">a\nACGT\n>zz",
    "comment_at_eof": ">a\nACGT\n>zz comment",
    "gt_inside_lines": ">a >x\nAC>GT\nA>\n>b\nCC\n",
    "blank_lines": ">a\n\n\nAC\n\n>b\n\n",
    "simulated": ">read=1,forward,position=5-15,length=10,chr1\nACGTACGTAC\n>read=2,reverse,position=7-11,length=4,chr2\nGGGG\n",
    # not eligible for the mapped path (CR, FASTQ, leading junk): both thread counts take the streaming reader
    "crlf": ">a\r\nAC\r\nGT\r\n>b\r\nTT\r\n",
    "fastq_mixed": ">a\nAC\n@q\nGGTT\n+\nIIII\n>b\nTT\n",
    "junk_before_header": "xx\n>a\nAC\n",
    "plus_line": ">a\nAC\n+\nGT\n>b\nTT\n",
}


@pytest.mark.parametrize("shape", sorted(FASTA_SHAPES))
def test_reads_loader_thread_count_does_not_matter(tmp_path, shape):
    f = tmp_path / "r.fa"
    f.write_bytes(FASTA_SHAPES[shape].encode())
    try:
        hostio.set_threads(1)
        want = _snapshot_reads(f)
        for t in (2, 3, 8):
            hostio.set_threads(t)
            assert hostio.get_threads() == t
            assert _snapshot_reads(f) == want, (shape, t)
    finally:
        hostio.set_threads(0)


def test_loaders_and_writers_thread_count_random(tmp_path):
    rng = np.random.default_rng(77)
    n = 700
    lens = rng.integers(0, 900, n)
    lens[rng.integers(0, n, 20)] = 0
    names = [f"read{i}/x" for i in range(n)]
    with open(tmp_path / "r.fa", "wb") as f:
        for i in range(n):
            seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(lens[i])))
            width = int(rng.choice([0, 60, 7]))
            f.write(b">" + names[i].encode() + (b" extra words" if i % 3 == 0 else b"") + b"\n")
            if width == 0:
                f.write(seq + b"\n")
            else:
                for k in range(0, len(seq), width):
                    f.write(seq[k:k + width] + b"\n")
    m = 5000
    q, t = rng.integers(0, n, m), rng.integers(0, n, m)
    cols = [lens.astype(np.int32), q, np.zeros(m, np.int64), np.zeros(m, np.int64), t, np.zeros(m, np.int64), np.zeros(m, np.int64)]
    for side, (ids, si, ei) in enumerate(((q, 2, 3), (t, 5, 6))):
        L = np.maximum(lens[ids], 1)
        a, b = rng.integers(0, L), rng.integers(0, L)
        cols[si], cols[ei] = np.minimum(a, b), np.minimum(np.maximum(a, b) + 1, lens[ids])
    keep = (lens[q] > 0) & (lens[t] > 0)
    cols = [cols[0]] + [c[keep].astype(np.int32) for c in cols[1:]]
    write_paf(tmp_path / "o.paf", names, *cols)
    p = RaftParams(est_cov=3, repeat_length=100, interval_length=100, read_length=300, overlap_length=50, flanking_length=20)
    outs = {}
    try:
        for th in (1, 5):
            hostio.set_threads(th)
            reads = hostio.Reads(str(tmp_path / "r.fa"))
            got = hostio.load_paf(str(tmp_path / "o.paf"), reads)
            for a, b in zip(got, cols[1:]):
                assert np.array_equal(a, b)
            for a, b in zip(hostio.load_paf(str(tmp_path / "o.paf"), reads, two_steps=True), cols[1:]):
                assert np.array_equal(a, b)
            assert reads.lengths.tolist() == lens.tolist()
            res = oracle_run(p, reads.lengths, *got)
            hostio.write_outputs(str(tmp_path / f"t{th}"), reads, p.reso, res)
            outs[th] = {s: open(tmp_path / f"t{th}.{s}", "rb").read()
                        for s in ("coverage.txt", "long_repeats.txt", "long_repeats.bed", "reads.fasta")}
            reads.close()
    finally:
        hostio.set_threads(0)
    assert outs[1] == outs[5]
    assert len(outs[1]["reads.fasta"]) > 0 and len(outs[1]["coverage.txt"]) > 0


def _bgzf(data: bytes, block: int = 30000) -> bytes:
    """`data` as a BGZF file (SAM spec 4.1): gzip members of at most 64 KiB with their size in a 'BC' extra field, + the empty end member."""
    import struct
    import zlib
    out = b""
    for k in list(range(0, len(data), block)) + [None]:
        chunk = b"" if k is None else data[k:k + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(comp) + 8
        out += (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + comp +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    return out


@pytest.mark.parametrize("threads", [1, 6])
def test_blocked_gzip_inputs_are_inflated_by_all_workers(tmp_path, threads):
    """Round 5 (VERDICT r04 "missing" 5): a BGZF file -- bgzip's output, the usual form of .fa.gz / .paf.gz next to htslib tools --
    is inflated member by member on all host threads; a plain .gz keeps the single stream.  Reads and records are the same from
    plain text, plain gzip and BGZF, with one thread (everything through the streaming reader) and with six."""
    p, cols, exp, meta = load_case("s300_default")
    names = [f"r{i}" for i in range(len(cols[0]))]
    write_fasta(tmp_path / "reads.fa", names, cols[0])
    write_paf(tmp_path / "o.paf", names, *cols)
    fa, paf = (tmp_path / "reads.fa").read_bytes(), (tmp_path / "o.paf").read_bytes()
    (tmp_path / "b.fa.gz").write_bytes(_bgzf(fa)); (tmp_path / "b.paf.gz").write_bytes(_bgzf(paf))
    (tmp_path / "g.fa.gz").write_bytes(gzip.compress(fa)); (tmp_path / "g.paf.gz").write_bytes(gzip.compress(paf))
    assert gzip.decompress((tmp_path / "b.fa.gz").read_bytes()) == fa              # (a BGZF file is a valid multi-member gzip file)
    broken = bytearray(_bgzf(paf)); broken[len(broken) // 2] ^= 0x55
    (tmp_path / "x.paf.gz").write_bytes(bytes(broken))
    try:
        hostio.set_threads(threads)
        want = None
        for r_name, p_name in (("reads.fa", "o.paf"), ("g.fa.gz", "g.paf.gz"), ("b.fa.gz", "b.paf.gz"), ("b.fa.gz", "o.paf")):
            reads = hostio.Reads(str(tmp_path / r_name))
            got = hostio.load_paf(str(tmp_path / p_name), reads)
            snap = (reads.lengths.tolist(), [reads.name(i) for i in (0, 1, reads.n - 1)], [reads.bases(i) for i in (0, reads.n // 2, reads.n - 1)], [g.tolist() for g in got])
            if want is None:
                want = snap
                for a, b in zip(got, cols[1:]):
                    assert np.array_equal(a, b)
            assert snap == want, (r_name, p_name)
            reads.close()
        # a damaged member: the blocked reader declines (crc / inflate error) and the single stream takes over -- an error or what zlib
        # salvages, never a crash
        reads = hostio.Reads(str(tmp_path / "reads.fa"))
        try:
            hostio.load_paf(str(tmp_path / "x.paf.gz"), reads)
        except hostio.HostError:
            pass
        reads.close()
    finally:
        hostio.set_threads(0)


# ---- split_naive (the reference's comparator tool) ----------------------------------------------------------------

SPLIT_INPUT = (">a first\nACGTACGTAC\nGTAC\n>empty\n>b\nTT\n>a\nCCCCCCCCCCCCCCCCCCCCCC\n"      # a repeated name, an empty read
               "@q1\nACGTACGTA\n+\nIIIIIIIII\n")
SPLIT_GOLDEN_7 = (">a_1\nACGTACG\n>a_2\nTACGTAC\n>b_1\nTT\n>a_1\nCCCCCCC\n>a_2\nCCCCCCC\n>a_3\nCCCCCCC\n>a_4\nC\n"
                  ">q1_1\nACGTACG\n>q1_2\nTA\n")


@pytest.mark.parametrize("threads", [1, 4])
def test_split_naive_matches_reference(tmp_path, threads):
    import subprocess
    src = tmp_path / "in.fa"
    src.write_text(SPLIT_INPUT)
    try:
        hostio.set_threads(threads)
        assert hostio.split_naive(str(src), str(tmp_path / "ours.fa"), 7) == 5
    finally:
        hostio.set_threads(0)
    ours = (tmp_path / "ours.fa").read_text()
    assert ours == SPLIT_GOLDEN_7                      # golden captured from the compiled reference tool
    ref = os.path.join(ROOT, "oracle", "_ref", "split_naive")
    if os.path.exists(ref):                            # and against the tool itself where it is built
        for L in (1, 7, 10, 1000):
            subprocess.run([ref, str(src), str(tmp_path / "ref.fa"), str(L)], check=True)
            hostio.split_naive(str(src), str(tmp_path / "ours.fa"), L)
            assert (tmp_path / "ours.fa").read_bytes() == (tmp_path / "ref.fa").read_bytes(), L
    cli = os.path.join(ROOT, "raft_amd", "bin", "split_naive")
    if os.path.exists(cli):
        r = subprocess.run([cli, str(src), str(tmp_path / "cli.fa"), "7"], stdout=subprocess.PIPE)
        assert r.returncode == 0 and (tmp_path / "cli.fa").read_text() == SPLIT_GOLDEN_7
        assert subprocess.run([cli, str(src)], stdout=subprocess.PIPE).returncode == 1     # usage
    with pytest.raises(hostio.HostError):
        hostio.split_naive(str(src), str(tmp_path / "x.fa"), 0)


def test_packed_coverage_roundtrip_and_writer(tmp_path):
    """The transfer encoding of cov[] (one byte per window + exceptions for values >= 255): the C++ decoder restores
    the array and the packed writer's bytes equal the int32 writer's, for 1, 3 and 8 threads."""
    rng = np.random.default_rng(5)
    nb = rng.integers(0, 4000, 300)
    off = np.zeros(len(nb) + 1, np.int64)
    np.cumsum(nb, out=off[1:])
    cov = rng.integers(0, 200, int(off[-1])).astype(np.int32)
    hot = rng.integers(0, cov.size, 4000)
    cov[hot] = rng.choice([254, 255, 256, 300, 70000, 2**31 - 1], hot.size)
    cov[:3] = [255, 0, 1000]; cov[-2:] = [255, 999]
    c8, xi, xv = hostio.pack_coverage(cov)
    assert c8.dtype == np.uint8 and (c8 == 255).sum() == xi.size and bool((np.diff(xi) > 0).all())
    assert np.array_equal(hostio.unpack_coverage(c8, xi, xv), cov)
    c16, yi, yv = hostio.pack_coverage(cov, width=2)      # two bytes per window: escape 65535
    assert c16.dtype == np.uint16 and (c16 == 65535).sum() == yi.size and 0 < yi.size < xi.size
    assert np.array_equal(hostio.unpack_coverage(c16, yi, yv), cov)
    with pytest.raises(hostio.HostError):                # an exception pointing at a window that is not 255
        hostio.unpack_coverage(c8, np.array([1], np.int64), np.array([7], np.int32))
    lib = hostio.load_library()
    for threads in (1, 3, 8):
        hostio.set_threads(threads)
        a, b = str(tmp_path / f"a{threads}.txt"), str(tmp_path / f"b{threads}.txt")
        assert lib.raft_host_write_coverage(a.encode(), len(nb), 50, off.ctypes.data, cov.ctypes.data) == 0
        hostio.write_coverage_packed(b, len(nb), 50, off, c8, xi, xv)
        assert open(a, "rb").read() == open(b, "rb").read()
        hostio.write_coverage_packed(b, len(nb), 50, off, c16, yi, yv)
        assert open(a, "rb").read() == open(b, "rb").read()
    hostio.set_threads(0)
    empty = np.empty(0, np.int64)
    assert hostio.unpack_coverage(np.empty(0, np.uint8), empty, np.empty(0, np.int32)).size == 0


@pytest.mark.parametrize("threads", [1, 4])
def test_paf_loader_reports_the_symmetric_flag(tmp_path, threads):
    """chop.hpp:171-184 inside the tokeniser: the flag equals the oracle's on the golden cases and on edge shapes."""
    hostio.set_threads(threads)
    try:
        for name in sorted(MAN["synthetic"]):
            p, cols, exp, meta = load_case(name)
            names = [f"r{i}" for i in range(len(cols[0]))]
            write_fasta(tmp_path / "r.fa", names, cols[0])
            write_paf(tmp_path / "o.paf", names, *cols)
            reads = hostio.Reads(str(tmp_path / "r.fa"))
            got, sym = hostio.load_paf(str(tmp_path / "o.paf"), reads, with_flag=True)
            assert sym == meta["symmetric"], name
        (tmp_path / "r.fa").write_text(">a\nACGTACGT\n>b\nACGTAC\n")
        reads = hostio.Reads(str(tmp_path / "r.fa"))
        line = lambda q, qs, qe, t, ts, te: f"{q}\t8\t{qs}\t{qe}\t+\t{t}\t6\t{ts}\t{te}\t1\t1\t60\n"
        cases = {
            "self-mirror record 0 alone is not a mirror": (line("a", 1, 5, "a", 1, 5), 0),
            "self-mirror repeated later": (line("a", 1, 5, "a", 1, 5) * 2, 1),
            "mirror after junk lines": ("junk\n" + line("a", 0, 4, "b", 1, 5) + "x\ty\n" + line("b", 1, 5, "a", 0, 4), 1),
            "coordinates differ": (line("a", 0, 4, "b", 1, 5) + line("b", 1, 5, "a", 0, 3), 0),
            "only record 0": (line("a", 0, 4, "b", 1, 5), 0),
        }
        for what, (text, want) in cases.items():
            (tmp_path / "o.paf").write_text(text)
            got, sym = hostio.load_paf(str(tmp_path / "o.paf"), reads, with_flag=True)
            rl = np.array([8, 6], np.int32)
            assert sym == want == oracle_run(RaftParams(est_cov=2), rl, *got)["symmetric"], what
    finally:
        hostio.set_threads(0)


def test_fastq_empty_sequence_consumes_its_quality_line(tmp_path):
    """kseq.h:290 reads one quality line in any case: after an empty-sequence FASTQ record its (empty) quality line is
    consumed and the next record parses; a quality line starting with '@' is not a header; a non-empty quality string
    after an empty sequence ends the file there (kseq returns -2), exactly as the compiled reference does."""
    import subprocess
    from raft_testlib import REF_BIN, have_ref_bin
    cases = {"empty_ok": "@a\nACGT\n+\n@@II\n@e\n\n+\n\n@b\nAC\n+\nII\n",
             "empty_then_garbage": "@a\nACGT\n+\nIIII\n@e\n\n+\n@x\n@b\nAC\n+\nII\n"}
    want = {"empty_ok": [("a", 4), ("e", 0), ("b", 2)], "empty_then_garbage": [("a", 4)]}
    for name, text in cases.items():
        fq = tmp_path / f"{name}.fq"
        fq.write_text(text)
        reads = hostio.Reads(str(fq))
        got = [(reads.name(i), int(reads.lengths[i])) for i in range(reads.n)]
        assert got == want[name], (name, got)
        if have_ref_bin():                               # and the reference itself: one coverage line per read it loaded
            (tmp_path / "o.paf").write_text("a\t4\t0\t4\t+\ta\t4\t0\t4\t1\t1\t1\n")
            r = subprocess.run([REF_BIN, "-e", "2", "-o", name, fq.name, "o.paf"], cwd=tmp_path, stdout=subprocess.PIPE)
            assert r.returncode == 0
            assert len(open(tmp_path / f"{name}.coverage.txt").read().splitlines()) == len(want[name]), name


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_group_offsets(threads):
    """raft_host_group_offsets: per-read record offsets of a query column that is a handful of sorted runs, against
    numpy's searchsorted; streams of another shape have no grouped form."""
    hostio.set_threads(threads)
    try:
        rng = np.random.default_rng(4)
        for n_reads, per_run, k in ((1, 5, 1), (700, 3000, 2), (70000, 600000, 2), (70000, 400000, 4), (50, 0, 1), (300, 40, 3)):
            runs = [np.sort(rng.integers(0, n_reads, per_run)).astype(np.int32) for _ in range(k)]
            if per_run and k > 1:
                for i in range(1, k):                      # make sure a new run really steps back
                    runs[i][0] = 0
                    runs[i - 1][-1] = n_reads - 1
            qid = np.concatenate(runs) if per_run else np.empty(0, np.int32)
            off = hostio.group_offsets(n_reads, qid, max_runs=4)
            k_eff = k if per_run else 1
            assert off is not None and off.shape == (k_eff, n_reads + 1), (n_reads, per_run, k)
            base = 0
            for i in range(k_eff):
                r = runs[i] if per_run else qid
                want = base + np.searchsorted(r, np.arange(n_reads + 1), side="left")
                assert np.array_equal(off[i], want), (n_reads, per_run, k, i)
                base += len(r)
            assert off[-1, -1] == qid.size
        q = np.sort(rng.integers(0, 100, 5000)).astype(np.int32)
        assert hostio.group_offsets(100, np.concatenate([q] * 5), max_runs=4) is None          # five runs
        assert hostio.group_offsets(100, np.concatenate([q] * 5), max_runs=5).shape == (5, 101)
        assert hostio.group_offsets(100, rng.permutation(q), max_runs=4) is None                 # shuffled
        bad = q.copy(); bad[-1] = 100
        assert hostio.group_offsets(100, bad) is None                                            # id out of range
        bad[-1] = 99; bad[0] = -1
        assert hostio.group_offsets(100, bad) is None
    finally:
        hostio.set_threads(0)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_pack_windows(threads):
    """raft_host_pack_windows against the window arithmetic of profileCoverage (repeat.hpp:69-72) in numpy: the windows an
    interval touches, 0 for one that touches none; negative coordinates and windows beyond 16 bits are reported."""
    hostio.set_threads(threads)
    try:
        rng = np.random.default_rng(9)
        for n, reso in ((0, 50), (1, 50), (1000, 1), (400000, 50), (400000, 7), (300000, 32767)):
            top = min(65535 * reso, 2**31 - 1)
            a = rng.integers(0, top, n).astype(np.int32)
            b = np.minimum(a.astype(np.int64) + rng.integers(-3 * reso, 20 * reso, n), top).clip(0).astype(np.int32)
            if n > 10:
                b[:5] = 0
                a[5:8] = 0; b[5:8] = 1
            w = hostio.pack_windows(a, b, reso)
            first = a // reso
            last1 = np.where(b > 0, (b.astype(np.int64) - 1) // reso + 1, 0)
            live = last1 > first
            want = np.where(live, first | (last1 << 16), 0).astype(np.uint32)
            assert w is not None and w.dtype == np.uint32 and np.array_equal(w, want), (n, reso)
        a = rng.integers(0, 10**6, 300000).astype(np.int32)
        b = a + 500
        for idx in ([299999], [12, 270000], [150000, 150001]):
            x = a.copy(); x[idx] = -1
            with pytest.raises(hostio.HostError) as e:
                hostio.pack_windows(x, b, 50)
            assert e.value.code == hostio.ERR_COORD and e.value.index == idx[0]
            y = b.copy(); y[idx] = -7
            with pytest.raises(hostio.HostError) as e:
                hostio.pack_windows(a, y, 50)
            assert e.value.code == hostio.ERR_COORD and e.value.index == idx[0]
        far = b.copy(); far[77777] = 65535 * 50 + 1
        assert hostio.pack_windows(a, far, 50) is None
        far[77777] = 65535 * 50
        assert hostio.pack_windows(a, far, 50)[77777] == (int(a[77777]) // 50) | (65535 << 16)
        far[3] = -1                                     # both kinds: the negative coordinate is the error
        with pytest.raises(hostio.HostError):
            hostio.pack_windows(a, far, 50)
        for reso in (0, -1, 32768):
            with pytest.raises(hostio.HostError) as e:
                hostio.pack_windows(a, b, reso)
            assert e.value.code == hostio.ERR_ARG
        out = np.empty(a.size + 10, np.uint32)
        w = hostio.pack_windows(a, b, 50, out=out)
        assert w.base is out or w.ctypes.data == out.ctypes.data
    finally:
        hostio.set_threads(0)


def encode_d4(cov, forced=()):
    """numpy restatement of the four-bit step encoding (include/raft_hip.h "delta4"); `forced`: windows escaped regardless."""
    cov = np.asarray(cov, np.int64)
    step = np.diff(np.concatenate([[0], cov]))
    code = np.where(np.abs(step) <= 7, step + 8, 0).astype(np.uint8)
    code[list(forced)] = 0
    exc = np.flatnonzero(code == 0)
    pad = np.concatenate([code, np.zeros(len(code) & 1, np.uint8)])
    nib = (pad[0::2] | (pad[1::2] << 4)).astype(np.uint8)
    anchor = np.concatenate([[0], cov[1023::1024]])[: (len(cov) + 1023) // 1024].astype(np.int32)
    return nib, anchor, exc.astype(np.int64), cov[exc].astype(np.int32)


@pytest.mark.parametrize("threads", [1, 4])
def test_delta4_decoder_and_writer(tmp_path, threads):
    """raft_host_unpack_coverage_d4 / raft_host_write_coverage_d4 against a numpy encoder: steps, escapes (large steps and
    forced ones), anchors at block starts, odd lengths; the text equals the int32 writer's."""
    hostio.set_threads(threads)
    try:
        rng = np.random.default_rng(31)
        for n in (0, 1, 2, 1023, 1024, 1025, 5000, 5_000_001):
            walk = rng.integers(-2, 3, n)
            jumps = rng.random(n) < 0.002
            walk[jumps] = rng.integers(-300, 300, int(jumps.sum()))
            cov = np.abs(np.cumsum(walk)).astype(np.int32)
            forced = rng.integers(0, n, min(n, 50)) if n else []
            nib, anchor, xi, xv = encode_d4(cov, forced)
            got = hostio.unpack_coverage_d4(n, nib, anchor, xi, xv)
            assert np.array_equal(got, cov), n
            if 0 < n <= 5000:
                lens = rng.multinomial(n, np.ones(7) / 7)
                off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                hostio.write_coverage_d4(str(tmp_path / "d4.txt"), 7, 50, off, nib, anchor, xi, xv)
                code, ei, ev = hostio.pack_coverage(cov, 2)
                hostio.write_coverage_packed(str(tmp_path / "ref.txt"), 7, 50, off, code, ei, ev)
                assert open(tmp_path / "d4.txt", "rb").read() == open(tmp_path / "ref.txt", "rb").read()
        # an escape without its entry, an entry on a window that is not escaped: rejected
        cov = np.arange(3000, dtype=np.int32) * 9
        nib, anchor, xi, xv = encode_d4(cov)
        with pytest.raises(hostio.HostError):
            hostio.unpack_coverage_d4(3000, nib, anchor, xi[:-1], xv[:-1])
        cov2 = np.zeros(3000, np.int32)
        nib, anchor, _, _ = encode_d4(cov2)
        with pytest.raises(hostio.HostError):
            hostio.unpack_coverage_d4(3000, nib, anchor, np.array([5], np.int64), np.array([1], np.int32))
    finally:
        hostio.set_threads(0)


# ---- the tokeniser's fast paths (round 6: sixteen bytes at a time, eight digits at a time, names in groups of eight, columns written in
# place) against a plain restatement of paf.hpp:50-87 + chop.hpp:157-160, on a text large enough for several workers ----------------

def _strtol_as_int32(field: str) -> int:
    """strtol(field, NULL, 10) -> uint32 -> int, as the reference converts a coordinate (chop.hpp:157-160)."""
    m = re.match(r"[ \t\n\v\f\r]*([+-]?)(\d*)", field)
    if not m or not m.group(2):
        return 0
    v = int(m.group(2))
    v = min(v, 2 ** 63 - 1 if m.group(1) != "-" else 2 ** 63)          # strtol saturates
    if m.group(1) == "-":
        v = -v
    v &= 0xffffffff
    return v - (1 << 32) if v >= (1 << 31) else v


@pytest.mark.parametrize("threads", [1, 2, 7])
def test_paf_tokeniser_fast_paths_against_the_rules(tmp_path, threads):
    rng = np.random.default_rng(20241008 + threads)
    n = 300
    names = [f"m64011_{i}/{int(rng.integers(1, 10 ** 6))}/ccs" if i % 3 else f"r{i}" for i in range(n)]
    with open(tmp_path / "r.fa", "w") as f:
        for nm in names:
            f.write(f">{nm}\nACGT\n")
    odd_numbers = ["", " 7", "+12", "-5", "12x", "x12", "00000000000000000042", "99999999999999999999", "4294967295", "2147483648", "123456789", "0"]
    lines, want = [], [[] for _ in range(6)]
    for k in range(60000):
        q, t = int(rng.integers(0, n)), int(rng.integers(0, n))
        if k % 5:                                     # (a PAF grouped by query: the line before's name, mostly)
            q = prev_q
        prev_q = q
        f4 = [str(int(rng.integers(0, 10 ** int(rng.integers(1, 9))))) for _ in range(4)]
        if k % 97 == 0:
            f4[int(rng.integers(0, 4))] = odd_numbers[int(rng.integers(0, len(odd_numbers)))]
        fields = [names[q], "100", f4[0], f4[1], "+", names[t], "100", f4[2], f4[3], "5", "6", "60"]
        kind = k % 211
        if kind == 0:
            fields = fields[:int(rng.integers(1, 10))]             # fewer than ten fields: no record (paf.hpp:84-85)
        elif kind == 1:
            fields = fields[:10]                                    # exactly ten
        elif kind == 2:
            fields += ["tp:A:P", "cm:i:9"]                          # more than eleven
        eol = "\r\n" if k % 389 == 0 else "\n"
        lines.append("\t".join(fields) + eol)
        if len(fields) >= 10:
            for c, v in zip(range(6), (q, _strtol_as_int32(f4[0]), _strtol_as_int32(f4[1]), t, _strtol_as_int32(f4[2]), _strtol_as_int32(f4[3]))):
                want[c].append(v)
    if threads == 2:
        lines[-1] = lines[-1].rstrip("\r\n")                        # a last line without its newline
    text = "".join(lines)
    assert len(text) > (1 << 21)                                    # (several workers: the loader uses one below 1 MB)
    (tmp_path / "o.paf").write_text(text)
    try:
        hostio.set_threads(threads)
        reads = hostio.Reads(str(tmp_path / "r.fa"))
        got = hostio.load_paf(str(tmp_path / "o.paf"), reads)
        for c in range(6):
            assert np.array_equal(got[c], np.asarray(want[c], np.int64).astype(np.int32)), c
        # an unknown name in the middle of the text: reported, whichever worker and group of lines meets it
        bad = lines[:40000] + ["nobody\t1\t2\t3\t+\t" + names[0] + "\t1\t2\t3\t4\t5\t6\n"] + lines[40000:]
        (tmp_path / "bad.paf").write_text("".join(bad))
        with pytest.raises(hostio.HostError) as e:
            hostio.load_paf(str(tmp_path / "bad.paf"), reads)
        assert e.value.code == hostio.ERR_UNKNOWN_NAME and e.value.what == "nobody"
        reads.close()
    finally:
        hostio.set_threads(0)
