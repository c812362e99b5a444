#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s6; mkdir -p $OUT
export RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_diag.so
for i in 1 2 3 4; do
  echo "## process $i columns" | tee -a $OUT/ctr.txt
  RAFT_VARIANT=5 PROBE_FORM=columns PROBE_WIDTH=4 timeout 600 python3 tools/mode_probe.py RAFT_WAVE_COUNTERS=8,16,24,32 2 2>&1 | grep "RAFT_WAVE" | tee -a $OUT/ctr.txt
done
echo "## windows" | tee -a $OUT/ctr.txt
RAFT_VARIANT=5 PROBE_FORM=windows PROBE_WIDTH=1 timeout 600 python3 tools/mode_probe.py RAFT_WAVE_COUNTERS=8,16,32 2 2>&1 | grep "RAFT_WAVE" | tee -a $OUT/ctr.txt
rocprofv3 -L 2>/dev/null | grep -i -B2 -A12 "pc.sampl" | head -60 > $OUT/pcs_list.txt
