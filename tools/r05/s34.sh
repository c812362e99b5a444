#!/bin/bash
# the final kernel with parts switched off (-DRAFT_WAVE_DIAG build, RAFT_WAVE_MODE bits: 8 no coverage stores, 4 no scatter, 2 no run scan),
# one process and one context per form; then SQ counters of the window-record form
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s34; mkdir -p $OUT
export RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_diag.so
for form in columns windows; do
  w=4; [ $form = windows ] && w=1
  echo "## $form, $w bytes per window out" | tee -a $OUT/modes.txt
  RAFT_NO_PLACEMENT_TRIAL=1 PROBE_FORM=$form PROBE_WIDTH=$w timeout 600 python3 tools/mode_probe.py RAFT_WAVE_MODE=0,2,8,10,12,14 2 2>&1 | grep "RAFT_WAVE_MODE" | tee -a $OUT/modes.txt
done
echo "## windows, four-bit steps out" | tee -a $OUT/modes.txt
RAFT_NO_PLACEMENT_TRIAL=1 PROBE_FORM=windows PROBE_WIDTH=8 timeout 600 python3 tools/mode_probe.py RAFT_WAVE_MODE=0,2,8,14 2 2>&1 | grep "RAFT_WAVE_MODE" | tee -a $OUT/modes.txt
unset RAFT_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab --input windows --cov-width 1"
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1_win -- python3 bench.py $B > $OUT/sq1_win.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/sq2_win -- python3 bench.py $B > $OUT/sq2_win.log 2>&1
python3 - <<'PY' | tee gpurun_out/s34/counters_win.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/s34/sq*_win")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pileup_wave_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()): print(d.split("/")[-1], k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY
find gpurun_out/s34 -name "*.csv" -size +2M -delete
