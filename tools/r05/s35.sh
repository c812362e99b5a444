#!/bin/bash
# tile_desc_kernel with an interpolated first probe; the kernel with parts switched off (diag build)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s35; mkdir -p $OUT
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_routed.py tests/test_gpu_grouped.py tests/test_gpu_consistency.py tests/test_gpu_windows.py tests/test_gpu_dist.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
tools/pass_timeline.sh s35_tl > $OUT/tl.txt 2>&1; head -7 $OUT/tl.txt | cut -c1-100
tools/pass_timeline.sh s35_tl8 --reads 412500 > $OUT/tl8.txt 2>&1; head -7 $OUT/tl8.txt | cut -c1-100
tools/pass_timeline.sh s35_tlul --workload ultralong > $OUT/tlul.txt 2>&1; head -7 $OUT/tlul.txt | cut -c1-100
RAFT_NO_PLACEMENT_TRIAL=1 timeout 600 python3 tools/r05/quantum_probe.py 0,7936,11904,15872 3 2>&1 | grep "quantum\|Error\|assert" | tee $OUT/quantum_full.txt
export RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_diag.so
for form in columns windows; do
  w=4; [ $form = windows ] && w=1
  echo "## $form, $w bytes per window out" | tee -a $OUT/modes.txt
  RAFT_NO_PLACEMENT_TRIAL=1 PROBE_FORM=$form PROBE_WIDTH=$w timeout 600 python3 tools/mode_probe.py RAFT_WAVE_MODE=0,2,8,10,12,14 2 2>&1 | grep "RAFT_WAVE_MODE" | tee -a $OUT/modes.txt
done
