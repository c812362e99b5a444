#!/bin/bash
# session 3: fused tail -- parity suite subset, bench, timelines
OUT=gpurun_out/r06_c; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo smoke rc=$?; tail -2 $OUT/smoke.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_wave.py tests/test_gpu_configs.py tests/test_gpu_grouped.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -5 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
python bench.py $B > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
python bench.py $B --reads 412500 --steps 20 --warmup 3 > $OUT/bench_slice.json 2> $OUT/bench_slice.err; echo slice rc=$?
tools/pass_timeline.sh r06_c_tl412 --reads 412500 > $OUT/pass_timeline_slice412k.txt 2>&1
tools/pass_timeline.sh r06_c_tl > $OUT/pass_timeline.txt 2>&1
python3 - <<'PY'
import json
for f in ("bench","bench_slice"):
    d=json.load(open(f"gpurun_out/r06_c/{f}.json")); r=d["roofline"]
    print(f, "ms/step",round(d["ms_per_step"],4),"kernel",round(r["kernel_ms"],4),"pass",round(r["pass_device_ms"],4),"frac",round(r["frac"],3),"pass_frac",round(r["pass_frac"],3))
PY
head -12 $OUT/pass_timeline_slice412k.txt | cut -c1-100; head -12 $OUT/pass_timeline.txt | cut -c1-100
