#!/bin/bash
# the int32 coverage stores write-through (sc1: the line leaves the L2) or non-temporal (nt) against plain: does the record
# over-read (4.7 GB fetched for 3.5 GB) go away when the stores stop turning the L2 over?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s36; mkdir -p $OUT
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4))"; }
B="--steps 12 --warmup 3 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
for i in 1 2 3; do for v in base sc1 nt; do
  RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_$v.so timeout 300 python3 bench.py $B 2>$OUT/err.txt | line "cols $v" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_$v.so timeout 300 python3 bench.py $B --workload ultralong 2>$OUT/err.txt | line "ul_cols $v" | tee -a $OUT/ab.txt
done; done
for v in base sc1; do
  RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_$v.so timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab > $OUT/fetch_$v.log 2>&1
  RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_$v.so timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab > $OUT/write_$v.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/s36/traffic.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/s36/fetch_*") + glob.glob("gpurun_out/s36/write_*")):
    if d.endswith(".log"): continue
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pileup_wave_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(d.split("/")[-1], k, "n=%d mean=%.6g KiB" % (len(v), sum(v)/len(v)))
PY
find gpurun_out/s36 -name "*.csv" -size +2M -delete
