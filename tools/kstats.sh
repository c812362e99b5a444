#!/bin/bash
# per-kernel average times of the engine's kernels over a few passes (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats; mkdir -p gpurun_out/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/kstats/log.txt 2>&1
python3 - <<'PY'
import csv, glob
tot = 0
for f in glob.glob("gpurun_out/kstats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "raft::" in n or "scan_" in n:
            per_pass = float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / 6
            tot += per_pass
            print("%-64s calls %3s avg_us %8.1f per_pass_us %8.1f" % (n.replace("void ", "").replace("raft::", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3, per_pass))
print("sum per pass (us): %.0f" % tot)
PY
tail -1 gpurun_out/kstats/log.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
