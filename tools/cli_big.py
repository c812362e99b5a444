#!/usr/bin/env python3
"""The `raft` CLI above toy size, text in -> text out, against the compiled reference on the same files (GPU box).

  python tools/cli_big.py [n_reads=500000] [est_cov=30]

tools/gen_set.cpp (built here with g++) writes SURVEY's S50k shape at ten times the reads: ~10 GB of FASTA, ~4.6e7 PAF
records in ~2.7 GB.  Both binaries run on those files; the four output files are compared by md5; the CLI's stage clock
(RAFT_TIMING=1) and the reference's wall time are printed.  Keep the output under profiles/."""
import hashlib
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
est_cov = int(sys.argv[2]) if len(sys.argv) > 2 else 30


def md5_of(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


need = n_reads * 20000 * 5 + (8 << 30)                    # inputs + two sets of outputs
cands = [d for d in (os.environ.get("RAFT_BIG_DIR"), "/dev/shm", "/tmp") if d and os.path.isdir(d)]
work = None
for d in cands:
    if shutil.disk_usage(d).free > need:
        work = os.path.join(d, f"raft_big_{os.getpid()}")
        break
if work is None:
    sys.exit(f"cli_big: no directory with {need >> 30} GB free among {cands}")
os.makedirs(work)
try:
    import tempfile
    gen = os.path.join(tempfile.mkdtemp(prefix="raft_gen_"), "gen_set")     # (/dev/shm is mounted noexec)
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "gen_set.cpp"), "-o", gen], check=True)
    fa, paf = os.path.join(work, "reads.fa"), os.path.join(work, "overlaps.paf")
    t0 = time.perf_counter()
    r = subprocess.run([gen, str(n_reads), "20000", "30", "20241008", fa, paf], stderr=subprocess.PIPE, check=True)
    print(f"work dir {work}; generator {time.perf_counter() - t0:.1f} s: {r.stderr.decode().strip()}")
    print(f"reads.fa {os.path.getsize(fa) / 1e9:.2f} GB, overlaps.paf {os.path.getsize(paf) / 1e9:.2f} GB")
    n_rec = int(r.stderr.decode().split()[-2])
    runs = {}
    for tag, exe, env in (("raft_amd (MI355X)", os.path.join(ROOT, "raft_amd", "bin", "raft"), {"RAFT_TIMING": "1"}),
                          ("raft_amd, second run (page cache warm)", os.path.join(ROOT, "raft_amd", "bin", "raft"), {"RAFT_TIMING": "1", "RAFT_PIPE_TRACE": "1"}),
                          ("reference (oracle/_ref/raft, 1 thread)", os.path.join(ROOT, "oracle", "_ref", "raft"), {})):
        if not os.path.exists(exe):
            print(f"{tag}: {exe} is missing -- skipped")
            continue
        out = os.path.join(work, "out_" + ("ref" if "reference" in tag else "gpu"))
        os.makedirs(out, exist_ok=True)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-e", str(est_cov), "-o", "x", fa, paf], cwd=out, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        wall = time.perf_counter() - t0
        t_end = time.time()
        print(f"\n== {tag}: exit {p.returncode}, wall {wall:.2f} s")
        for ln in p.stderr.decode().splitlines():
            if ln.startswith("PIPE"):
                print("   " + ln)
            if ln.startswith("TIMING leaving-at"):
                # what lies behind main(): the kernel tearing the process down (mappings, page-locked ranges, the GPU context)
                print(f"   TIMING {'teardown':16s} {t_end - float(ln.split()[2]):8.3f} s   (from main() leaving to the parent seeing the exit)")
            elif ln.startswith("TIMING"):
                print("   " + ln)
                if ln.startswith("TIMING engine+fetch"):
                    sec = float(ln.split()[2])
                    print(f"   -> engine+fetch: {n_rec / sec:.3e} PAF records/s (upload, passes and download of every chunk overlapped)")
        for ln in p.stdout.decode().splitlines():
            if ln.startswith(("INFO, Symmetric", "INFO, length", "high_cov", "coverage per window", "fraction_of")):
                print("   " + ln)
        if p.returncode != 0:
            print(p.stdout.decode()[-2000:], p.stderr.decode()[-2000:])
        runs[tag] = {f: md5_of(os.path.join(out, f)) for f in sorted(os.listdir(out))}
        runs[tag + " wall"] = wall
    tags = [t for t in runs if not t.endswith(" wall")]
    if len(tags) >= 2:
        a, b = runs[tags[0]], runs[tags[-1]]
        print("\n== outputs")
        ok = True
        for f in sorted(set(a) | set(b)):
            same = a.get(f) == b.get(f)
            ok = ok and same
            print(f"   {f:24s} {a.get(f)}  {'==' if same else '!='}  {b.get(f)}")
        print("   four files byte-identical to the reference:" if "reference" in tags[-1] else "   runs agree:", ok)
        if "reference" in tags[-1]:
            print(f"   wall: reference {runs[tags[-1] + ' wall']:.2f} s / raft_amd {runs[tags[1] + ' wall']:.2f} s = "
                  f"{runs[tags[-1] + ' wall'] / runs[tags[1] + ' wall']:.1f}x")
finally:
    shutil.rmtree(work, ignore_errors=True)
