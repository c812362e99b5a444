#!/bin/bash
# the switches of the paths this round touched: the whole GPU suite with the runtime's waits instead of the stamped lines, a subset
# without the chunk mapping
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s38; mkdir -p $OUT
RAFT_NO_SPIN=1 timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed" | sed 's/^/RAFT_NO_SPIN=1: /' | tee $OUT/summary.txt
RAFT_NO_VMM=1 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_windows.py -x -q 2>&1 | grep "passed\|failed" | sed 's/^/RAFT_NO_VMM=1: /' | tee -a $OUT/summary.txt
