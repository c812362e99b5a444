#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s11; mkdir -p $OUT
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4))"; }
for i in 1 2 3; do for lib in raft_amd/lib/libraft_hip_base.so raft_amd/lib/libraft_hip.so; do
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab 2>$OUT/err.txt | line "cols $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 8 --warmup 2 --cov-width 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab 2>$OUT/err.txt | line "cols_w1 $lib" | tee -a $OUT/ab.txt
done; done
RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip.so timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
