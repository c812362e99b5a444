#!/bin/bash
OUT=gpurun_out/r06_f; mkdir -p $OUT
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
for env in "X=1" "RAFT_SYNC_BEFORE_PILE=1" "RAFT_NO_SPECULATE=1" "X=2" "RAFT_SYNC_BEFORE_PILE=1" "RAFT_NO_SPECULATE=1"; do
  env $env python bench.py $B --reads 412500 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$env', 'ms/step',round(d['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),'pass',round(r['pass_device_ms'],4))"
done
