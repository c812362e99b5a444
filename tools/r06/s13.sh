#!/bin/bash
OUT=gpurun_out/r06_l; mkdir -p $OUT
python tools/cli_big.py 500000 > $OUT/cli_s500k.txt 2>&1; echo rc=$?
grep -v "^   PIPE" $OUT/cli_s500k.txt | head -70
