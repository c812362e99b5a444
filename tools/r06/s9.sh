#!/bin/bash
OUT=gpurun_out/r06_i; mkdir -p $OUT
python tools/pipe_trace.py 3300000 0 columns_d4 2> $OUT/pipe_columns_d4.txt
python tools/pipe_trace.py 3300000 0 windows_d4 2> $OUT/pipe_windows_d4.txt
tail -70 $OUT/pipe_columns_d4.txt
