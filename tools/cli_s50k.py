#!/usr/bin/env python3
"""End-to-end CLI timing on the survey's S50k-like set (50 k reads, ~4.3 M PAF records, 1 GB FASTA):
raft_amd/bin/raft vs the compiled reference (oracle/_ref/raft, when present), outputs compared by md5."""
import hashlib, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raft_testlib import write_fasta, REF_BIN
from raft_amd.synth import make_overlaps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
o = make_overlaps(n, seed=2, device="cuda:0" if len(sys.argv) > 2 else "cpu")
cols = [c.cpu().numpy() for c in (o.read_len,) + o.columns()]
d = tempfile.mkdtemp(dir="/tmp")
names = [f"r{i}" for i in range(n)]
t0 = time.time()
write_fasta(os.path.join(d, "reads.fa"), names, cols[0])
rl = cols[0]
with open(os.path.join(d, "overlaps.paf"), "w") as f:      # vectorised PAF writer
    q, qs, qe, t, ts, te = cols[1:]
    lines = np.char.add(np.char.add("r", q.astype(str)), "\t")
    for a in (rl[q], qs, qe):
        lines = np.char.add(np.char.add(lines, a.astype(str)), "\t")
    lines = np.char.add(lines, "+\tr")
    lines = np.char.add(np.char.add(lines, t.astype(str)), "\t")
    for a in (rl[t], ts, te, qe - qs, qe - qs):
        lines = np.char.add(np.char.add(lines, a.astype(str)), "\t")
    lines = np.char.add(lines, "60\n")
    f.write("".join(lines.tolist()))
print(f"inputs written in {time.time()-t0:.1f} s: {os.path.getsize(d+'/reads.fa')/1e6:.0f} MB FASTA, {os.path.getsize(d+'/overlaps.paf')/1e6:.0f} MB PAF, {len(cols[1])} records")

def run(exe, prefix, fa="reads.fa", paf="overlaps.paf", env=None):
    t = time.time()
    r = subprocess.run([exe, "-e", "30", "-o", prefix, fa, paf], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, RAFT_TIMING="1", **(env or {})))
    dt = time.time() - t
    print("".join(l + "\n" for l in r.stdout.decode().splitlines() if l.startswith("TIMING")), end="")
    md = {x: hashlib.md5(open(os.path.join(d, f"{prefix}.{x}"), "rb").read()).hexdigest() for x in ("reads.fasta", "coverage.txt", "long_repeats.txt", "long_repeats.bed")} if r.returncode == 0 else {}
    return r.returncode, dt, md, r.stdout.decode()[-300:]

ours = os.path.join(ROOT, "raft_amd", "bin", "raft")
for rep in range(2):
    rc, dt, md_o, tail = run(ours, "ours")
    print(f"raft (MI355X engine) run {rep}: rc={rc} wall {dt:.2f} s -> {len(cols[1])/dt:.3e} PAF records/s end-to-end (text in, text out)")
if os.path.exists(REF_BIN):
    rc, dt, md_r, tail = run(REF_BIN, "ref")
    print(f"reference raft (1 thread): rc={rc} wall {dt:.2f} s -> {len(cols[1])/dt:.3e} PAF records/s")
    print("outputs identical:", md_o == md_r, md_o)
# the reference's own quick-start input shape: gz FASTA (chop.hpp:93), and a gz PAF (paf.hpp:29)
t0 = time.time()
subprocess.run("gzip -1 -k reads.fa overlaps.paf", shell=True, cwd=d, check=True)
print(f"gzip -1 of both inputs: {time.time()-t0:.1f} s ({os.path.getsize(d+'/reads.fa.gz')/1e6:.0f} MB + {os.path.getsize(d+'/overlaps.paf.gz')/1e6:.0f} MB)")
for label, env in (("gz inputs, helper inflater thread", None), ("gz inputs, RAFT_HOST_THREADS=1 (sequential reader)", {"RAFT_HOST_THREADS": "1"})):
    rc, dt, md_g, tail = run(ours, "oursgz", "reads.fa.gz", "overlaps.paf.gz", env)
    print(f"raft (MI355X engine), {label}: rc={rc} wall {dt:.2f} s -> {len(cols[1])/dt:.3e} PAF records/s; outputs identical to the plain run: {md_g == md_o}")
if os.path.exists(REF_BIN):
    rc, dt, md_rg, tail = run(REF_BIN, "refgz", "reads.fa.gz", "overlaps.paf.gz")
    print(f"reference raft (1 thread), gz inputs: rc={rc} wall {dt:.2f} s; outputs identical: {md_rg == md_o}")
subprocess.run(["rm", "-rf", d])
