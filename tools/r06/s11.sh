#!/bin/bash
OUT=gpurun_out/r06_k; mkdir -p $OUT
python -m pytest tests/test_gpu_routed.py tests/test_gpu_parity.py tests/test_gpu_exchange.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -4 $OUT/pytest.log
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "full_size or config1 or fuzz" >> $OUT/pytest.log 2>&1; echo pytest2 rc=$?; tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
for w in shuffle nonsym; do
python bench.py $B --$w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo $w rc=$?
python3 -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']
print('$w ms/step',round(d['ms_per_step'],3),'kernel',round(r['kernel_ms'],3),'frac',round(r['frac'],3),'pass',round(r['pass_device_ms'],3),'pass_frac',round(r['pass_frac'],3))"
done
tools/pass_timeline.sh r06_k_tlsh --shuffle 2>&1 | grep -v "at::native" | head -30 | cut -c1-110
