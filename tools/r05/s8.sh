#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s8; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 600 python3 bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python3 -c "import json; d=json.load(open('$OUT/bench.json')); print('default ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'pass_frac', d['roofline'].get('pass_frac')); print('placement_ab', d.get('placement_ab')); print('e2e', d.get('e2e',{}).get('records_per_s'))"
for lib in raft_amd/lib/libraft_hip.so raft_amd/lib/libraft_hip_c6w3.so raft_amd/lib/libraft_hip_c8w3.so; do
  RAFT_HIP_LIB=$PWD/$lib timeout 600 python3 bench.py --workload ultralong --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab 2>$OUT/err.txt | tail -1 > $OUT/ul.json
  python3 -c "import json; d=json.load(open('$OUT/ul.json')); print('ultralong $lib ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'pass', d['roofline']['pass_device_ms'])"
done
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --shuffle 2>$OUT/err.txt | tail -1 > $OUT/sh.json
python3 -c "import json; d=json.load(open('$OUT/sh.json')); print('shuffle ms/step', d['ms_per_step'], 'pass', d['roofline']['pass_device_ms'])"
