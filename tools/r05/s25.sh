#!/bin/bash
# the fixed part of a pass, second try: finer sample index (256 k), joint sample bisection, per-read scalars loaded first in
# finalize_count/fill, no scratch in finalize_count, unrolled totals, fast divisions in the geometry scan
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s25; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_routed.py tests/test_gpu_grouped.py tests/test_gpu_windows.py tests/test_gpu_consistency.py -x -q 2>&1 | grep "passed\|failed\|Error" | tee $OUT/pytest.txt
tools/pass_timeline.sh s25_tl > $OUT/tl.txt 2>&1; head -13 $OUT/tl.txt | cut -c1-100
tools/pass_timeline.sh s25_tl8 --reads 412500 > $OUT/tl8.txt 2>&1; head -13 $OUT/tl8.txt | cut -c1-100
tools/pass_timeline.sh s25_tlul --workload ultralong > $OUT/tlul.txt 2>&1; head -13 $OUT/tlul.txt | cut -c1-100
B="--steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4))"; }
for i in 1 2; do
python3 bench.py $B 2>$OUT/err.txt | line full | tee -a $OUT/ab.txt
python3 bench.py $B --reads 412500 2>$OUT/err.txt | line eighth | tee -a $OUT/ab.txt
python3 bench.py $B --reads 50000 2>$OUT/err.txt | line r50k | tee -a $OUT/ab.txt
done
python3 bench.py $B --shuffle 2>$OUT/err.txt | line shuffle | tee -a $OUT/ab.txt
python3 bench.py $B --nonsym 2>$OUT/err.txt | line nonsym | tee -a $OUT/ab.txt
tools/pass_timeline.sh s25_tlsh --shuffle > $OUT/tlsh.txt 2>&1; head -24 $OUT/tlsh.txt | cut -c1-100
