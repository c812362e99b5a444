#!/bin/bash
# the suite's speculation / parity tests with the round's switches thrown
mkdir -p gpurun_out/r06_end
for e in RAFT_NO_KEEP_GEOMETRY=1 RAFT_NO_SPECULATE=1 RAFT_GRADED_QUANTUM=0; do
  env $e timeout 900 python3 -m pytest tests -m gpu -x -q -k "speculate or parity or consistency" > gpurun_out/r06_end/pytest_$e.txt 2>&1; echo "$e rc=$?"; tail -1 gpurun_out/r06_end/pytest_$e.txt
done
