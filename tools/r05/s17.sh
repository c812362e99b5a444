#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s17; mkdir -p $OUT
B="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
for lib in raft_amd/lib/libraft_hip.so raft_amd/lib/libraft_hip_rs8.so; do
  RAFT_HIP_LIB=$PWD/$lib timeout 600 python3 bench.py $B --shuffle 2>$OUT/err.txt | tail -1 > $OUT/b.json
  python3 -c "import json; d=json.load(open('$OUT/b.json')); print('$lib shuffle ms/step', round(d['ms_per_step'],3), 'pass', round(d['roofline']['pass_device_ms'],3), d['self_check']['sum_cov_equals_windows_touched'])" | tee -a $OUT/ab.txt
done
RAFT_LIBRARY_SORT=1 timeout 600 python3 bench.py $B --shuffle 2>$OUT/err.txt | tail -1 > $OUT/b.json
python3 -c "import json; d=json.load(open('$OUT/b.json')); print('library shuffle ms/step', round(d['ms_per_step'],3), 'pass', round(d['roofline']['pass_device_ms'],3))" | tee -a $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in raft_amd/lib/libraft_hip.so raft_amd/lib/libraft_hip_rs8.so; do
RAFT_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$(basename $lib) -- python3 bench.py $B --shuffle > $OUT/stats.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/s17/stats.txt
import csv, glob
for f in sorted(glob.glob("gpurun_out/s17/stats_*/**/*kernel_stats.csv", recursive=True)):
    print(f.split('/')[2])
    for r in csv.DictReader(open(f)):
        if "rs_" in r["Name"] or "expand" in r["Name"] or "unzip" in r["Name"]:
            print("  stats", r["Name"][:60], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
find $OUT -name "*.csv" -size +1M -delete
timeout 900 python3 -m pytest tests/test_gpu_routed.py tests/test_gpu_exchange.py -x -q 2>&1 | grep "passed\|failed" | tee -a $OUT/ab.txt
