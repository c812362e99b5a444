#!/bin/bash
# (a) the stream-group kernel against the build before it, in one process; (b) the CLI on the 10 GB set with the new tokeniser
mkdir -p gpurun_out/r06_s18
O=gpurun_out/r06_s18
A=raft_amd/lib/libraft_hip.so; B=raft_amd/lib/libraft_hip_old.so
python3 tools/lib_ab.py $A $B 3300000 3 columns 4 2>&1 | tail -2
python3 tools/lib_ab.py $B $A 3300000 3 columns 4 2>&1 | tail -2
python3 tools/lib_ab.py $A $B 412500 3 columns 4 2>&1 | tail -2
python3 tools/lib_ab.py $A $B 3300000 3 windows 1 2>&1 | tail -2
python3 tools/cli_big.py 500000 > $O/cli_s500k.txt 2>&1; echo "cli rc=$?"
grep -E "wall|TIMING|identical" $O/cli_s500k.txt | head -60
grep -c PIPE $O/cli_s500k.txt
