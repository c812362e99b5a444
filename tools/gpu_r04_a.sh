#!/bin/bash
# round 4, first look at the wave kernel: golden parity under RAFT_VARIANT=5, then old vs new on the bench set
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_a
export RAFT_VARIANT=5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r04_a/parity_v5.txt 2>&1; echo "parity v5 rc $?"
tail -15 gpurun_out/r04_a/parity_v5.txt
unset RAFT_VARIANT
for v in 0 5; do
  timeout 600 python bench.py --variant $v --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-packed-leg --no-six-column-leg > gpurun_out/r04_a/bench_v$v.json 2> gpurun_out/r04_a/bench_v$v.err
  echo "bench v$v rc $?"; tail -3 gpurun_out/r04_a/bench_v$v.err
  python -c "import json;d=json.loads([l for l in open('gpurun_out/r04_a/bench_v$v.json') if l.startswith('{')][0]);r=d['roofline'];print('v$v ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4))"
done
