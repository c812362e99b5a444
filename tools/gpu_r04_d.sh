#!/bin/bash
# round 4: streaming wave kernel -- benches of the three workloads in the headline and product forms, timeline, then the GPU suites
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_d
line() { python - "$1" "$2" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]);r=d['roofline']
    print(tag,'ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4),'pass_frac',round(r['pass_frac'],4))
except Exception as e: print(tag,'no line',e)
PY
}
for w in hg002 ultralong s50k; do
  timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-six-column-leg --no-packed-leg > gpurun_out/r04_d/bench_$w.json 2> gpurun_out/r04_d/bench_$w.err; line gpurun_out/r04_d/bench_$w.json "$w columns/int32"
  timeout 600 python bench.py --workload $w --input windows --cov-width 1 --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-six-column-leg --no-packed-leg > gpurun_out/r04_d/bench_${w}_w1.json 2> gpurun_out/r04_d/bench_${w}_w1.err; line gpurun_out/r04_d/bench_${w}_w1.json "$w windows/byte"
done
tools/pass_timeline.sh r04_tl | head -12
tools/pass_timeline.sh r04_tl_s50k --workload s50k | head -12
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04_d/gpu_suite.txt 2>&1; echo "gpu suite rc $?"
tail -8 gpurun_out/r04_d/gpu_suite.txt
