#!/bin/bash
# where does the CLI's wall clock go that its stage clock does not see?  (process start, process exit)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s22; mkdir -p $OUT
W=/dev/shm/raft_sweep_$$; mkdir -p $W
G=$(mktemp -d)/gen_set
g++ -O2 -std=c++17 tools/gen_set.cpp -o $G
$G 500000 20000 30 20241008 $W/reads.fa $W/overlaps.paf 2> $OUT/gen.txt
run() { tag=$1; shift
  mkdir -p $W/o; cd $W/o
  s=$(date +%s.%N)
  env "$@" RAFT_TIMING=1 $GRAFT_REPO_ROOT/raft_amd/bin/raft -e 30 -o x $W/reads.fa $W/overlaps.paf > /dev/null 2> $W/err.txt
  e=$(date +%s.%N)
  python3 -c "
import re
t=[float(x) for x in re.findall(r'TIMING \S+(?: \(rest\))?\s+([0-9.]+) s', open('$W/err.txt').read())]
print('$tag wall %.2f stages %.2f unaccounted %.2f' % ($e-$s, sum(t), $e-$s-sum(t)))" | tee -a $GRAFT_REPO_ROOT/$OUT/exit.txt
  cd $GRAFT_REPO_ROOT; rm -rf $W/o
}
run default A=1
run default2 A=1
run no_pin RAFT_NO_PIN=1
run no_warm_up RAFT_NO_WARM_UP=1
run clean_exit RAFT_CLEAN_EXIT=1
run no_vmm RAFT_NO_VMM=1
rm -rf $W
