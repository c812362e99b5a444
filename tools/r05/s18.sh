#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s18; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_routed.py tests/test_gpu_exchange.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 | tee $OUT/pytest.txt
B="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
for f in "--shuffle" "--nonsym"; do for nw in 0 1; do
  if [ $nw = 1 ]; then export RAFT_NO_BUCKET_WINDOWS=1; else unset RAFT_NO_BUCKET_WINDOWS; fi
  timeout 600 python3 bench.py $B $f 2>$OUT/err.txt | tail -1 > $OUT/b.json
  python3 -c "import json; d=json.load(open('$OUT/b.json')); print('$f coordinate_pairs=$nw', 'ms/step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'pass', round(d['roofline']['pass_device_ms'],3), 'pass_frac', round(d['roofline'].get('pass_frac',0),4), all(d['self_check'].values()))" | tee -a $OUT/ab.txt
done; done
unset RAFT_NO_BUCKET_WINDOWS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shuffle -- python3 bench.py $B --shuffle > $OUT/stats_shuffle.log 2>&1
python3 - <<'PY' | tee gpurun_out/s18/stats.txt
import csv, glob
for f in glob.glob("gpurun_out/s18/stats_shuffle/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "at::" in r["Name"] or "rocclr" in r["Name"] or "rocprim" in r["Name"]: continue
        print("stats", r["Name"][:70], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
find $OUT -name "*.csv" -size +1M -delete
