#!/bin/bash
# final build: 25 runs of the tests that caught the unstamped hand-over, plus the smoke entry
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s41; mkdir -p $OUT
fails=0
for i in $(seq 1 25); do
  timeout 300 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $OUT/run.txt 2>&1
  if ! grep -q " passed" $OUT/run.txt || grep -q "failed" $OUT/run.txt; then fails=$((fails+1)); tail -60 $OUT/run.txt > $OUT/fail_$i.txt; fi
done
echo "final build: $fails failures of 25" | tee $OUT/summary.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $OUT/summary.txt
