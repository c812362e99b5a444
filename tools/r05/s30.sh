#!/bin/bash
# finalize_count / finalize_fill with a register path for up to four repeats: parity, then the pass's timeline
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s30; mkdir -p $OUT
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_wave.py tests/test_gpu_windows.py tests/test_gpu_routed.py tests/test_gpu_grouped.py tests/test_gpu_consistency.py tests/test_gpu_cli.py tests/test_gpu_exchange.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
tools/pass_timeline.sh s30_tl > $OUT/tl.txt 2>&1; head -13 $OUT/tl.txt | cut -c1-100
tools/pass_timeline.sh s30_tl8 --reads 412500 > $OUT/tl8.txt 2>&1; head -13 $OUT/tl8.txt | cut -c1-100
tools/pass_timeline.sh s30_tlul --workload ultralong > $OUT/tlul.txt 2>&1; head -13 $OUT/tlul.txt | cut -c1-100
timeout 900 python3 tools/full_compare.py --workload hg002 > $OUT/full_compare_hg002.txt 2>&1; tail -3 $OUT/full_compare_hg002.txt
timeout 900 python3 tools/full_compare.py --workload ultralong > $OUT/full_compare_ultralong.txt 2>&1; tail -3 $OUT/full_compare_ultralong.txt
