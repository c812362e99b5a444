#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s7; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_routed.py tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_delta4.py tests/test_gpu_packed_output.py tests/test_gpu_wave.py -x -q 2>&1 | tail -25 | tee $OUT/pytest_subset.txt
for f in "--shuffle" "--nonsym"; do
  timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg $f 2>$OUT/err.txt | tail -1 > $OUT/bench$f.json
  python3 -c "import json,sys; d=json.load(open('$OUT/bench$f.json')); print('$f', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'pass', d['roofline']['pass_device_ms'], 'pass_frac', d['roofline'].get('pass_frac'), d['self_check'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shuffle -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --shuffle > $OUT/stats_shuffle.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/s7/stats_shuffle/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "at::" in r["Name"] or "rocclr" in r["Name"]: continue
        print("stats", r["Name"][:70], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
find $OUT -name "*.csv" -size +1M -delete
