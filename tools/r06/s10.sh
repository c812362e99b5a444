#!/bin/bash
OUT=gpurun_out/r06_j; mkdir -p $OUT
nproc; lscpu | grep -i "model name\|^CPU(s)\|Thread\|Socket\|NUMA node(s)"
for t in 8 16 24 32 48 64; do
  RAFT_DERIVE_THREADS=$t python tools/pipe_trace.py 3300000 0 columns_d4 2>&1 | grep "^pass" | tr '\n' ' ' | sed "s/^/derive threads $t: /"; echo
done
for ch in 8 16 24 32; do
  RAFT_DERIVE_THREADS=32 python tools/pipe_trace.py 3300000 $ch columns_d4 2>&1 | grep "^pass" | tr '\n' ' ' | sed "s/^/chunks $ch (32 thr): /"; echo
done
