#!/bin/bash
# session 1: does the library with hidden symbols run; baseline numbers of this round's first lease
OUT=gpurun_out/r06_a; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo smoke rc=$?
python -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -3 $OUT/pytest.log
python bench.py --no-e2e --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
python bench.py --reads 412500 --no-e2e --no-cpu-baseline --no-packed-leg --no-six-column-leg --no-placement-ab --steps 20 --warmup 3 > $OUT/bench_slice.json 2> $OUT/bench_slice.err; echo slice rc=$?
tools/pass_timeline.sh r06_a_tl412 --reads 412500 > $OUT/pass_timeline_slice412k.txt 2>&1
tools/pass_timeline.sh r06_a_tl > $OUT/pass_timeline.txt 2>&1
