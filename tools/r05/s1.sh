#!/bin/bash
# round 5 session 1: A/B of builds (separate processes, two rounds), then a PC-sampling attempt on the headline command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
OUT=gpurun_out/s1
line() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4), d['self_check'].get('sum_cov_equals_windows_touched'))"; }
for i in 1 2; do for lib in raft_amd/lib/libraft_hip_r04.so raft_amd/lib/libraft_hip.so raft_amd/lib/libraft_hip_s3072w4.so raft_amd/lib/libraft_hip_s3072w5.so raft_amd/lib/libraft_hip_s3072w6.so; do
  RAFT_BENCH_ABLATION=1 RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg 2>$OUT/err.txt | line "cols $lib" >> $OUT/ab.txt
  RAFT_BENCH_ABLATION=1 RAFT_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 8 --warmup 2 --input windows --cov-width 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg 2>$OUT/err.txt | line "win1 $lib" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
for m in stochastic host_trap; do
  timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $([ $m = stochastic ] && echo cycles || echo time) --pc-sampling-interval $([ $m = stochastic ] && echo 1048576 || echo 100) --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/pcs_$m -o pcs --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-windows-leg > $GRAFT_REPO_ROOT/$OUT/pcs_$m.log 2>&1
  echo "pcs $m rc=$?"; tail -5 $GRAFT_REPO_ROOT/$OUT/pcs_$m.log
  ls -la $GRAFT_REPO_ROOT/$OUT/pcs_$m 2>/dev/null | head
  if ls $GRAFT_REPO_ROOT/$OUT/pcs_$m/*/*pc_sampling* >/dev/null 2>&1 || ls $GRAFT_REPO_ROOT/$OUT/pcs_$m/*pc_sampling* >/dev/null 2>&1; then break; fi
done
python3 $GRAFT_REPO_ROOT/tools/r05/pcs_agg.py $GRAFT_REPO_ROOT/$OUT
du -sh $GRAFT_REPO_ROOT/$OUT/*
find $GRAFT_REPO_ROOT/$OUT -size +20M -delete
