#!/bin/bash
# tile_desc_kernel: the wave's searches through the sample index together (span in LDS)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s33; mkdir -p $OUT
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_routed.py tests/test_gpu_grouped.py tests/test_gpu_consistency.py tests/test_gpu_windows.py tests/test_gpu_dist.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
tools/pass_timeline.sh s33_tl > $OUT/tl.txt 2>&1; head -7 $OUT/tl.txt | cut -c1-100
tools/pass_timeline.sh s33_tl8 --reads 412500 > $OUT/tl8.txt 2>&1; head -7 $OUT/tl8.txt | cut -c1-100
tools/pass_timeline.sh s33_tlul --workload ultralong > $OUT/tlul.txt 2>&1; head -7 $OUT/tlul.txt | cut -c1-100
RAFT_NO_PLACEMENT_TRIAL=1 timeout 600 python3 tools/r05/quantum_probe.py 0,3968,7936,11904,15872 3 2>&1 | grep "quantum\|Error\|assert" | tee $OUT/quantum_full.txt
PROBE_READS=412500 RAFT_NO_PLACEMENT_TRIAL=1 timeout 600 python3 tools/r05/quantum_probe.py 0,3968,7936,15872 3 2>&1 | grep "quantum\|Error\|assert" | tee $OUT/quantum_eighth.txt
