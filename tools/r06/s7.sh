#!/bin/bash
# session 7: whole GPU suite on the fused head / tail, bench full + eighth in one lease
OUT=gpurun_out/r06_g; mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -6 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
python bench.py $B > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
python bench.py $B --reads 412500 --steps 20 --warmup 3 > $OUT/bench_slice.json 2> $OUT/bench_slice.err; echo slice rc=$?
python3 - <<'PY'
import json
for f in ("bench","bench_slice"):
    try:
        d=json.load(open(f"gpurun_out/r06_g/{f}.json")); r=d["roofline"]
        print(f, "ms/step",round(d["ms_per_step"],4),"kernel",round(r["kernel_ms"],4),"pass",round(r["pass_device_ms"],4),"frac",round(r["frac"],3),"pass_frac",round(r["pass_frac"],3))
    except Exception as e: print(f, "failed", e)
PY
