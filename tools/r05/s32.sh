#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s32; mkdir -p $OUT
RAFT_NO_PLACEMENT_TRIAL=1 timeout 600 python3 tools/r05/quantum_probe.py 0,7936,15872,23808,31744,47616,63488 3 2>&1 | grep "quantum\|Error\|assert" | tee $OUT/full.txt
PROBE_READS=412500 RAFT_NO_PLACEMENT_TRIAL=1 timeout 600 python3 tools/r05/quantum_probe.py 0,3968,7936,15872,31744 3 2>&1 | grep "quantum\|Error\|assert" | tee $OUT/eighth.txt
