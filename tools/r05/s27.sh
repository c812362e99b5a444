#!/bin/bash
# is the one failure of s25 reproducible, and is it the sizes hand-over seen without the runtime's wait?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s27; mkdir -p $OUT
for mode in spin nospin; do
  fails=0
  for i in $(seq 1 25); do
    if [ $mode = nospin ]; then export RAFT_NO_SPIN=1; else unset RAFT_NO_SPIN; fi
    timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $OUT/run.txt 2>&1
    if ! grep -q " passed" $OUT/run.txt || grep -q "failed" $OUT/run.txt; then fails=$((fails+1)); grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $OUT/run.txt | tail -70 > $OUT/fail_${mode}_$i.txt; fi
  done
  echo "$mode: $fails failures of 25" | tee -a $OUT/summary.txt
done
