#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s14
for sp in 8 1; do echo "## RAFT_VMM_SPREAD=$sp" | tee -a gpurun_out/s14/corr.txt; RAFT_VMM_SPREAD=$sp timeout 900 python3 tools/r05/placement_corr.py 7 2>&1 | grep context | tee -a gpurun_out/s14/corr.txt; done
