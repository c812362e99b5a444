#!/bin/bash
# kernel time of the pileup kernel vs resident workgroups per CU (latency-bound or resource-bound?)
for v in ${VARIANTS:-0 2}; do for b in 1 2 3 4 5; do
  RAFT_PILEUP_WG_PER_CU=$b timeout 300 python bench.py --reads ${READS:-2000000} --steps 4 --warmup 1 --no-cpu-baseline --variant $v ${EXTRA} > gpurun_out/occ.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/occ.log") if x.startswith("{")]
j=json.loads(l[-1]); print("variant $v wg/CU $b kernel_ms %.3f frac %.3f" % (j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
PY
done; done
