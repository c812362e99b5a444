#!/bin/bash
# host-thread sweep of the CLI on the 10 GB set (stage clock only)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s10; mkdir -p $OUT
W=/dev/shm/raft_sweep_$$; mkdir -p $W
G=$(mktemp -d)/gen_set
g++ -O2 -std=c++17 tools/gen_set.cpp -o $G
$G 500000 20000 30 20241008 $W/reads.fa $W/overlaps.paf 2> $OUT/gen.txt
for th in 0 16 32 64 128; do
  mkdir -p $W/o$th; cd $W/o$th
  echo "== RAFT_HOST_THREADS=$th" | tee -a $GRAFT_REPO_ROOT/$OUT/sweep.txt
  ( time RAFT_TIMING=1 RAFT_HOST_THREADS=$th $GRAFT_REPO_ROOT/raft_amd/bin/raft -e 30 -o x $W/reads.fa $W/overlaps.paf > /dev/null ) 2>&1 | grep "TIMING\|real" | grep -v "0.000 s" | tee -a $GRAFT_REPO_ROOT/$OUT/sweep.txt
  cd $GRAFT_REPO_ROOT; rm -rf $W/o$th
done
rm -rf $W
