#!/bin/bash
# A/B of wave-kernel builds in one GPU session: tools/ab_wave.sh <bench args>; every raft_amd/lib/libraft_hip_*.so, two rounds
cd $GRAFT_REPO_ROOT
for i in 1 2; do for lib in raft_amd/lib/libraft_hip_*.so; do
  RAFT_BENCH_ABLATION=1 RAFT_HIP_LIB=$PWD/$lib timeout 300 python bench.py --variant 5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$lib', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4), d['self_check'].get('sum_cov_equals_windows_touched'))"
done; done
