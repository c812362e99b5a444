#!/bin/bash
# wave kernel with fewer vector instructions (buffer loads for the record slots, scalar row offsets in the stores, bias folded into the
# start value, table entries by one ds_read2, error flags as scalar masks) against the build before (libraft_hip_base.so)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s28; mkdir -p $OUT
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_windows.py tests/test_gpu_delta4.py tests/test_gpu_routed.py tests/test_gpu_grouped.py tests/test_gpu_consistency.py tests/test_gpu_packed_output.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4))"; }
B="--steps 12 --warmup 3 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
for i in 1 2 3; do for lib in raft_amd/lib/libraft_hip_base.so raft_amd/lib/libraft_hip.so; do
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B 2>$OUT/err.txt | line "cols $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib RAFT_NO_PLACEMENT_TRIAL=1 timeout 300 python3 bench.py $B 2>$OUT/err.txt | line "cols_notrial $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B --input windows --cov-width 1 2>$OUT/err.txt | line "win_w1 $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B --input windows --cov-width 8 2>$OUT/err.txt | line "win_d4 $lib" | tee -a $OUT/ab.txt
  RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py $B --workload ultralong 2>$OUT/err.txt | line "ul_cols $lib" | tee -a $OUT/ab.txt
done; done
# ... and the hand-over in stamped lines: 30 runs of the tests that caught the unstamped form (s27: 1 of 25)
fails=0
for i in $(seq 1 30); do
  timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $OUT/run.txt 2>&1
  if ! grep -q " passed" $OUT/run.txt || grep -q "failed" $OUT/run.txt; then fails=$((fails+1)); grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $OUT/run.txt | tail -70 > $OUT/fail_$i.txt; fi
done
echo "stamped lines: $fails failures of 30" | tee -a $OUT/summary.txt
