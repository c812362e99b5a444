#!/bin/bash
# where does the one-shot stall of the host-to-host pipeline (a D2H hipMemcpyAsync that holds its caller ~16 ms) come from?
mkdir -p gpurun_out/r06_stall
O=gpurun_out/r06_stall
run() { tag=$1; shift; env "$@" python3 tools/pipe_trace.py 3300000 0 columns_d4 2> $O/trace_$tag.txt; grep -E "held its caller|^pass " $O/trace_$tag.txt | sed "s/^/[$tag] /"; }
run base X=1
run sigpool ROC_SIGNAL_POOL_SIZE=256
run nointr HSA_ENABLE_INTERRUPT=0
run derive4 RAFT_DERIVE_THREADS=4
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000000
run base2 X=1
