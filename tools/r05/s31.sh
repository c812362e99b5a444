#!/bin/bash
# the small kernels of a pass under the counters: what is finalize_count / finalize_fill / tile_desc's time made of?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s31; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_consistency.py -x -q 2>&1 | grep "passed\|failed" | tee $OUT/pytest.txt
tools/pass_timeline.sh s31_tl > $OUT/tl.txt 2>&1; head -13 $OUT/tl.txt | cut -c1-100
tools/pass_timeline.sh s31_tlul --workload ultralong > $OUT/tlul.txt 2>&1; sed -n 8,12p $OUT/tlul.txt | cut -c1-100
B="--steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg --no-placement-ab"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $B > $OUT/$name.log 2>&1; echo "pass $name rc=$?"; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
run tcc1 FETCH_SIZE WRITE_SIZE
run tcc2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
run tcc3 TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_WRITE_sum
run tcp1 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum
python3 - <<'PY' | tee gpurun_out/s31/counters.txt
import csv, glob, collections
for name in ("sq1","sq2","tcc1","tcc2","tcc3","tcp1"):
    files = glob.glob(f"gpurun_out/s31/{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            for tag in ("finalize_count", "finalize_fill", "tile_desc", "tile_first", "totals_kernel", "pileup_wave"):
                if tag in kn: agg[(tag, r["Counter_Name"])].append(float(r["Counter_Value"]))
            if "scan_apply" in kn: agg[("scan_apply_" + ("prep" if "ReadPrep" in kn else "count"), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{name:5s} {k[0]:18s} {k[1]:24s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
