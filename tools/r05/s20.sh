#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_b; mkdir -p $OUT
timeout 1200 python3 tools/full_compare.py --workload hg002 > $OUT/full_compare_hg002.txt 2>&1; tail -4 $OUT/full_compare_hg002.txt
timeout 1200 python3 tools/full_compare.py --workload ultralong > $OUT/full_compare_ultralong.txt 2>&1; tail -4 $OUT/full_compare_ultralong.txt
