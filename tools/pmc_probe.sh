#!/bin/bash
# PMC counter passes for the pileup kernel (rocprofv3 --pmc only; never combined with trace domains).
# usage: tools/pmc_probe.sh <reads> [bench args]      (summary on stdout: keep it under profiles/)
R=${1:-3300000}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
EXTRA=("$@")
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python3 bench.py --reads $R --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab "${EXTRA[@]}" > gpurun_out/pmc/$name.log 2>&1
  echo "pass $name rc=$?"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/stats -- python3 bench.py --reads $R --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab "${EXTRA[@]}" > gpurun_out/pmc/stats.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/pmc/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("stats", r["Name"][:60], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3), "pct", r["Percentage"])
PY
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections
for name in ("sq1","sq2","tcc1","tcc2"):
    files = glob.glob(f"gpurun_out/pmc/{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "pileup" in kn:
                tag = "wave" if "pileup_wave" in kn else "deep"
                agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{name:5s} {k[0]:8s} {k[1]:24s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
