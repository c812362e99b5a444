#!/bin/bash
# session 4: what the tail's parts cost (diagnostic switches)
OUT=gpurun_out/r06_d; mkdir -p $OUT
export RAFT_BENCH_ABLATION=1
for m in 0 1 2 3 4 7; do
  RAFT_TAIL_MODE=$m tools/pass_timeline.sh r06_d_m$m --reads 412500 2>&1 | grep "finalize" | cut -c1-60 | sed "s/^/mode $m eighth /"
  RAFT_TAIL_MODE=$m tools/pass_timeline.sh r06_d_M$m 2>&1 | grep "finalize" | cut -c1-60 | sed "s/^/mode $m full   /"
done
