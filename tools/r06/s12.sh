#!/bin/bash
OUT=gpurun_out/r06_k; mkdir -p $OUT
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
tools/pass_timeline.sh r06_k_tlsh --shuffle 2>&1 | grep "rs_\|unzip\|pileup_wave" | cut -c1-110
for w in shuffle nonsym; do
python bench.py $B --$w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo $w rc=$?
python3 -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']
print('$w ms/step',round(d['ms_per_step'],3),'kernel',round(r['kernel_ms'],3),'frac',round(r['frac'],3),'pass',round(r['pass_device_ms'],3),'pass_frac',round(r['pass_frac'],3))"
done
