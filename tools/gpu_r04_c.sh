#!/bin/bash
# round 4: the whole GPU suite with the wave kernel as the default; benches of the three workloads, variant 0 beside the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_c
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04_c/gpu_suite.txt 2>&1; echo "gpu suite rc $?"
tail -12 gpurun_out/r04_c/gpu_suite.txt
for w in hg002 ultralong s50k; do for v in 0 5; do
  timeout 600 python bench.py --workload $w --variant $v --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-six-column-leg --no-packed-leg > gpurun_out/r04_c/bench_${w}_v$v.json 2> gpurun_out/r04_c/bench_${w}_v$v.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04_c/bench_${w}_v$v.json') if l.startswith('{')][0]);r=d['roofline']
    print('$w v$v: ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4), 'pass_frac', round(r['pass_frac'],4))
except Exception as e: print('$w v$v no line', e, open('gpurun_out/r04_c/bench_${w}_v$v.err').read()[-600:])
PY
done; done
