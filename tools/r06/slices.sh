#!/bin/bash
# the eighth of the set, three more processes of the full bench line (legs included)
mkdir -p gpurun_out/r06_end
for i in 1 2 3; do python3 bench.py --reads 412500 > gpurun_out/r06_end/bench_slice412k_$i.json 2> gpurun_out/r06_end/bench_slice412k_$i.err; python3 - gpurun_out/r06_end/bench_slice412k_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("slice412k ms/step %.4f kernel %.4f pass %.4f" % (d["ms_per_step"], r["kernel_ms"], r["pass_device_ms"]))
PY
done
