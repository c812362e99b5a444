#!/bin/bash
# bench.py over a list of tile quanta (windows): prints quantum, ms/step, pileup kernel ms, roofline fraction
# usage: tools/sweep_q.sh "<q1 q2 ...>" [extra bench.py args]
QS=${1:-"4096"}; shift
for q in $QS; do
  python3 bench.py --no-cpu-baseline --tile-bins $q "$@" 2>/dev/null | tail -1 > /tmp/sweep_q.json
  python3 - "$q" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep_q.json"))
print("Q", sys.argv[1], "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
done
