#!/bin/bash
# non-temporal coverage stores in every form, against plain ones (libraft_hip_base.so)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s37; mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_windows.py tests/test_gpu_delta4.py tests/test_gpu_packed_output.py tests/test_gpu_consistency.py -x -q 2>&1 | grep "passed\|failed" | tee $OUT/pytest.txt
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4))"; }
B="--steps 12 --warmup 3 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
for i in 1 2 3; do for lib in raft_amd/lib/libraft_hip_base.so raft_amd/lib/libraft_hip.so; do
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B --input windows --cov-width 1 2>$OUT/err.txt | line "win_w1 $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B --input windows --cov-width 8 2>$OUT/err.txt | line "win_d4 $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B --cov-width 1 2>$OUT/err.txt | line "cols_w1 $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B 2>$OUT/err.txt | line "cols $lib" | tee -a $OUT/ab.txt
done; done
