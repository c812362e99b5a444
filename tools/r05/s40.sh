#!/bin/bash
# SQ counters of the final pileup kernel, headline form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s40; mkdir -p $OUT
B="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -- python3 bench.py $B > $OUT/sq1.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- python3 bench.py $B > $OUT/sq2.log 2>&1
python3 - <<'PY' | tee gpurun_out/s40/counters.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/s40/sq?")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pileup_wave_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(d.split("/")[-1], k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY
find gpurun_out/s40 -name "*.csv" -size +2M -delete
