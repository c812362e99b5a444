#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s2; mkdir -p $OUT
RAFT_VARIANT=6 timeout 900 python3 -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py tests/test_gpu_windows.py tests/test_gpu_delta4.py -x -q 2>&1 | tail -25 > $OUT/pytest_subset.txt
cat $OUT/pytest_subset.txt
for f in "columns 4" "columns 1" "windows 1" "windows 8"; do set -- $f
  PROBE_FORM=$1 PROBE_WIDTH=$2 timeout 600 python3 tools/r05/variant_probe.py 5,6 3 2>&1 | grep -v "^$" | tail -4 | tee -a $OUT/probe.txt
done
