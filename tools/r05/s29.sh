#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s29; mkdir -p $OUT
for v in new NOBIAS NOSOFF NOBUF; do
  export RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_$v.so
  timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep "passed\|failed" | tail -1 | sed "s/^/$v parity: /" | tee -a $OUT/summary.txt
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab 2>&1 | grep -o "self-check failed.*\|\"ms_per_step\": [0-9.]*" | head -1 | sed "s/^/$v bench: /" | tee -a $OUT/summary.txt
done
