#!/bin/bash
OUT=gpurun_out/r06_n; mkdir -p $OUT
python -m pytest tests/test_gpu_speculate.py tests/test_gpu_deep.py -x -q -m gpu -k "not parity_suites" > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -5 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
for rep in 1 2; do
python bench.py $B --reads 412500 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('eighth ms/step',round(d['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),'pass',round(r['pass_device_ms'],4))"
done
