#!/bin/bash
# final build: the speculation / hand-over tests many times over, then the whole suite and smoke() once more
mkdir -p gpurun_out/r06_end
fails=0
for i in $(seq 1 15); do
  timeout 600 python3 -m pytest tests/test_gpu_speculate.py tests/test_gpu_consistency.py -m gpu -x -q > gpurun_out/r06_end/stress_$i.txt 2>&1 || { fails=$((fails+1)); echo "run $i failed"; tail -5 gpurun_out/r06_end/stress_$i.txt; }
done
echo "stress: 15 runs, $fails failed"; tail -1 gpurun_out/r06_end/stress_15.txt
timeout 1500 python3 -m pytest tests -q -m gpu > gpurun_out/r06_end/pytest_gpu_last.txt 2>&1; echo "suite rc=$?"; tail -1 gpurun_out/r06_end/pytest_gpu_last.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_end/smoke_last.txt 2>&1; echo "smoke rc=$?"
