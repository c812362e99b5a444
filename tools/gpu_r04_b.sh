#!/bin/bash
# round 4: the wave kernel (variant 5) through the GPU suites, then old vs new on the bench set in three forms
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_b
for v in 0 5; do
  timeout 600 python bench.py --variant $v --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-six-column-leg > gpurun_out/r04_b/bench_v$v.json 2> gpurun_out/r04_b/bench_v$v.err
  echo "bench v$v rc $?"; tail -2 gpurun_out/r04_b/bench_v$v.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04_b/bench_v$v.json') if l.startswith('{')][0]);r=d['roofline']
    print('v$v six-column: ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4), 'pass_frac', round(r['pass_frac'],4))
    for k in ('packed_output','window_records','window_records_delta4'):
        if k in d: print('   ',k,'kernel_ms',round(d[k]['kernel_ms'],4),'pass_ms',round(d[k]['pass_device_ms'],4), 'ok', d[k]['equals_int32_pass'])
except Exception as e: print('no line', e)
PY
done
export RAFT_VARIANT=5
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_windows.py tests/test_gpu_delta4.py tests/test_gpu_grouped.py tests/test_gpu_packed_output.py tests/test_gpu_consistency.py -x -q > gpurun_out/r04_b/suites_v5.txt 2>&1; echo "suites v5 rc $?"
tail -25 gpurun_out/r04_b/suites_v5.txt
