#!/bin/bash
A=raft_amd/lib/libraft_hip.so; B=raft_amd/lib/libraft_hip_nt.so
python tools/lib_ab.py $A $B 3300000 3 columns 4 2>&1 | tail -2
python tools/lib_ab.py $B $A 3300000 3 columns 4 2>&1 | tail -2
python tools/lib_ab.py $A $B 3300000 3 windows 1 2>&1 | tail -2
