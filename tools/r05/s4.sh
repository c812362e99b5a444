#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s4; mkdir -p $OUT
RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_L3584w5.so RAFT_VARIANT=6 timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py tests/test_gpu_windows.py tests/test_gpu_delta4.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_subset.txt
for lib in raft_amd/lib/libraft_hip_L4096w4.so raft_amd/lib/libraft_hip_L3584w5.so raft_amd/lib/libraft_hip_L3072w6.so; do
  for f in "columns 4" "windows 1"; do set -- $f
    echo "## $lib" | tee -a $OUT/probe.txt
    RAFT_HIP_LIB=$PWD/$lib PROBE_FORM=$1 PROBE_WIDTH=$2 timeout 600 python3 tools/r05/variant_probe.py 5,6 2 2>&1 | grep -v "^$" | grep "^#\|variant\|Error\|error" | tee -a $OUT/probe.txt
  done
done
