#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s21; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_wave.py -x -q 2>&1 | grep "passed\|failed\|Error" | tee -a $OUT/pytest.txt
timeout 600 python3 bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python3 -c "
import json; d=json.load(open('$OUT/bench.json'))
print('ms/step', round(d['ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), 'pass_frac', round(d['roofline']['pass_frac'],3))
print('trial', d.get('placement_trial'))
print('ab', {k:round(v['kernel_ms'],3) for k,v in d.get('placement_ab',{}).items() if k!='note'})" | tee -a $OUT/trial.txt
