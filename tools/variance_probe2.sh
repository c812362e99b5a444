#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4), d['self_check'].get('sum_cov_equals_windows_touched'))"; }
for rep in 1 2; do for c in 0 1; do
run RAFT_CONTIG=$c
run RAFT_CONTIG=$c RAFT_WAVE_SHARE=0 RAFT_WAVE_BATCH=4
run RAFT_CONTIG=$c RAFT_WAVE_SHARE=0 RAFT_WAVE_BATCH=3
run RAFT_CONTIG=$c RAFT_WAVE_SHARE=3
run RAFT_CONTIG=$c RAFT_WAVE_PERM=619
run RAFT_CONTIG=$c RAFT_WAVE_PERM=1
done; done
