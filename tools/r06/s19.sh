#!/bin/bash
# the eighth of the set: how long do the kernel's last draws cost?  uniform quanta and graded ones
mkdir -p gpurun_out/r06_s19
O=gpurun_out/r06_s19
one() { tag=$1; shift; env "$@" python3 bench.py --reads 412500 --no-extra-legs --steps 40 --warmup 5 $EXTRA > $O/b_$tag.json 2> $O/b_$tag.err; python3 - $O/b_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("%-14s ms/step %.4f kernel %.4f pass %.4f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["pass_device_ms"]))
PY
}
for rep in 1 2; do
one base X=1
EXTRA="--tile-bins 3968" one q1 X=1
EXTRA="--tile-bins 5888" one q1.5 X=1
EXTRA="--tile-bins 11904" one q3 X=1
EXTRA= one g4_2_75 RAFT_GRADED=4,2,0.75,8
EXTRA= one g4_1_80 RAFT_GRADED=4,1,0.8,8
EXTRA= one g6_2_80 RAFT_GRADED=6,2,0.8,8
EXTRA= one g3_2_60 RAFT_GRADED=3,2,0.6,8
done
# smaller LDS arrays, five waves per SIMD for the window-record kernels
A=raft_amd/lib/libraft_hip.so
for B in raft_amd/lib/libraft_hip_s3072w5.so raft_amd/lib/libraft_hip_s3072w4.so; do
  echo "== $B"
  python3 tools/lib_ab.py $A $B 3300000 3 windows 1 2>&1 | tail -2
  python3 tools/lib_ab.py $A $B 3300000 3 windows 8 2>&1 | tail -2
  python3 tools/lib_ab.py $A $B 3300000 3 columns 4 2>&1 | tail -2
done
python3 tools/cli_big.py 500000 > $O/cli_s500k.txt 2>&1; echo "cli rc=$?"
grep -E "wall|TIMING (paf_load|engine|main|teardown|process)|identical" $O/cli_s500k.txt | head -30
