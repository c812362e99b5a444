#!/bin/bash
# Per-dispatch timeline of the engine's passes (rocprofv3 --kernel-trace): start/end of every kernel of the last bench step,
# relative to the step's first kernel, with the hardware queue each ran on.  usage: tools/pass_timeline.sh <tag> [bench args]
TAG=${1:-r03_tl}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab "$@" > gpurun_out/$TAG/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$TAG/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass: from the last ReadPrepLoader partials kernel on
idx = [i for i, r in enumerate(rows) if "scan_partials_kernel" in r["Kernel_Name"] and "ReadPrep" in r["Kernel_Name"]]
start = idx[-1]
g = [i for i, r in enumerate(rows) if "guess_runs_kernel" in r["Kernel_Name"]]
if g and g[-1] < start and start - g[-1] < 4: start = g[-1]     # (speculative pass: the guess starts beside the scans)
t0 = int(rows[start]["Start_Timestamp"])
out = open("gpurun_out/$TAG/timeline.txt", "w")
for r in rows[start:]:
    line = f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f} us  q{r.get("Queue_Id","?"):>3}  {r["Kernel_Name"][:90]}'
    print(line); out.write(line + "\n")
PY
