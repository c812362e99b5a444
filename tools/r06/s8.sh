#!/bin/bash
OUT=gpurun_out/r06_h; mkdir -p $OUT
python -m pytest tests/test_gpu_deep.py tests/test_gpu_wave.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -25 $OUT/pytest.log
