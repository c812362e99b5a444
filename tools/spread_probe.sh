#!/bin/bash
# kernel time of the default bench pass in several processes, per spread factor (RAFT_VMM_SPREAD)
cd $GRAFT_REPO_ROOT
for sp in "$@"; do
  echo -n "spread $sp:"
  for i in 1 2 3 4 5 6; do
    RAFT_VMM_SPREAD=$sp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(' %.3f' % d['roofline']['kernel_ms'], end='')"
  done; echo
done
