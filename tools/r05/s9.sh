#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s9; mkdir -p $OUT
nproc > $OUT/nproc.txt
timeout 1200 python3 tools/cli_big.py 500000 30 2>&1 | grep -v "PIPE chunk" | tee $OUT/cli_s500k.txt | tail -60
