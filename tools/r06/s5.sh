#!/bin/bash
# session 5: fused head + speculation: parity, bench, timelines
OUT=gpurun_out/r06_e; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo smoke rc=$?; tail -1 $OUT/smoke.log | cut -c1-80
python -m pytest tests/test_gpu_speculate.py tests/test_gpu_parity.py tests/test_gpu_wave.py tests/test_gpu_configs.py tests/test_gpu_grouped.py tests/test_gpu_windows.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -8 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
python bench.py $B > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
python bench.py $B --reads 412500 --steps 20 --warmup 3 > $OUT/bench_slice.json 2> $OUT/bench_slice.err; echo slice rc=$?
RAFT_NO_SPECULATE=1 python bench.py $B > $OUT/bench_nospec.json 2> $OUT/bench_nospec.err
RAFT_NO_SPECULATE=1 python bench.py $B --reads 412500 --steps 20 --warmup 3 > $OUT/bench_slice_nospec.json 2> $OUT/bench_slice_nospec.err
tools/pass_timeline.sh r06_e_tl412 --reads 412500 > $OUT/pass_timeline_slice412k.txt 2>&1
tools/pass_timeline.sh r06_e_tl > $OUT/pass_timeline.txt 2>&1
python3 - <<'PY'
import json
for f in ("bench","bench_nospec","bench_slice","bench_slice_nospec"):
    try:
        d=json.load(open(f"gpurun_out/r06_e/{f}.json")); r=d["roofline"]
        print(f, "ms/step",round(d["ms_per_step"],4),"kernel",round(r["kernel_ms"],4),"pass",round(r["pass_device_ms"],4),"frac",round(r["frac"],3),"pass_frac",round(r["pass_frac"],3))
    except Exception as e: print(f, "failed", e)
PY
grep -v "at::native" $OUT/pass_timeline_slice412k.txt | head -12 | cut -c1-100; grep -v "at::native" $OUT/pass_timeline.txt | head -12 | cut -c1-100
