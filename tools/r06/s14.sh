#!/bin/bash
OUT=gpurun_out/r06_m; mkdir -p $OUT
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "full_size or config4 or ultralong or config5" > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
for rep in 1 2 3; do
for g in 1 0; do
RAFT_GRADED_QUANTUM=$g python bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('graded=$g ms/step',round(d['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),'pass',round(r['pass_device_ms'],4),'fixed',round(r['pass_device_ms']-r['kernel_ms'],4))"
done; done
for g in 1 0; do
RAFT_GRADED_QUANTUM=$g python bench.py $B --workload ultralong 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('ultralong graded=$g ms/step',round(d['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),'pass',round(r['pass_device_ms'],4))"
done
tools/pass_timeline.sh r06_m_tl 2>&1 | grep -v "at::native\|rocclr" | head -10 | cut -c1-100
