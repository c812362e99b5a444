#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s15
rocm-smi --showmemvendor --showvbios 2>&1 | grep -v "^=\|^$" | head -5 >> gpurun_out/s15/corr.txt
timeout 900 python3 tools/r05/placement_corr.py 9 8,0,1 2>&1 | grep context | tee -a gpurun_out/s15/corr.txt
