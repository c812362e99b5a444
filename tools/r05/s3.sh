#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s3; mkdir -p $OUT
for lib in raft_amd/lib/libraft_hip_r4096w4.so raft_amd/lib/libraft_hip_r3072w4.so raft_amd/lib/libraft_hip_r3072w5.so; do
  for f in "columns 4" "windows 1"; do set -- $f
    echo "## $lib" | tee -a $OUT/probe.txt
    RAFT_HIP_LIB=$PWD/$lib PROBE_FORM=$1 PROBE_WIDTH=$2 timeout 600 python3 tools/r05/variant_probe.py 5,6 2 2>&1 | grep -v "^$" | grep "^#\|variant" | tee -a $OUT/probe.txt
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 5 6; do for form in columns windows; do
  extra=""; [ $form = windows ] && extra="--input windows --cov-width 1"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1_${v}_$form -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --variant $v $extra > $OUT/sq1_${v}_$form.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/sq2_${v}_$form -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --variant $v $extra > $OUT/sq2_${v}_$form.log 2>&1
done; done
python3 - <<'PY' | tee gpurun_out/s3/counters.txt
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/s3/sq*_*")):
    if not d.endswith(("columns","windows")): continue
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pileup_wave_kernel" in r["Kernel_Name"] or "pileup_ring_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(d.split("/")[-1], k, "n=%d mean=%.4g" % (len(v), sum(v)/len(v)))
PY
find gpurun_out/s3 -name "*.csv" -size +2M -delete
