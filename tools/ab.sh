#!/bin/bash
# A/B of two engine builds in ONE session (same device, interleaved): raft_amd/lib/libraft_hip_prev.so vs libraft_hip.so
for rep in 1 2 3; do for lib in prev cur; do
  if [ $lib = prev ]; then export RAFT_HIP_LIB=$GRAFT_REPO_ROOT/raft_amd/lib/libraft_hip_prev.so; else unset RAFT_HIP_LIB; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --variant ${VARIANT:-0} > gpurun_out/ab.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/ab.log") if x.startswith("{")]
j=json.loads(l[-1]); print("$lib rep $rep: ms/step %.3f kernel_ms %.3f" % (j["ms_per_step"], j["roofline"]["kernel_ms"]))
PY
done; done
