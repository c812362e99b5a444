#!/bin/bash
mkdir -p gpurun_out/r06_stall
O=gpurun_out/r06_stall
run() { tag=$1; shift; env RAFT_TRACE_PASSES=10 "$@" python3 tools/pipe_trace.py 3300000 0 columns_d4 2> $O/trace_$tag.txt; grep -E "held its caller|^pass " $O/trace_$tag.txt | sed "s/^/[$tag] /"; }
run many X=1
run nosdma HSA_ENABLE_SDMA=0
