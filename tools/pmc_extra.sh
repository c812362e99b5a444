#!/bin/bash
# More PMC passes for the pileup kernel: in-flight levels (latency = LEVEL / INSTS), instruction fetch, LDS / VMEM waits.
# usage: tools/pmc_extra.sh      (rocprofv3 --pmc only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcx
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcx/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg > gpurun_out/pmcx/$name.log 2>&1
  echo "pass $name rc=$?"; }
run a SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM
run c SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_FLAT
# (a pass with TCP_* / TA_* derived counters hung rocprofv3 on this pool for its whole time limit: left out)
python3 - <<'PY'
import csv, glob, collections
for name in "abc":
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcx/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "pileup_fast" in kn:
                targs = kn.split("<", 1)[1].split(">", 1)[0].split(",")
                tag = "extra" if len(targs) > 5 and targs[5].strip() in ("true", "1") else "regular"
                agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{name} {k[0]:8s} {k[1]:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
tail -3 gpurun_out/pmcx/*.log | grep -i "error\|invalid\|not" | head
