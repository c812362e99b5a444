#!/bin/bash
# the small sets' quantum alone (ranges drawn ahead measured slower and dropped), against the build before
mkdir -p gpurun_out/r06_s22
O=gpurun_out/r06_s22
A=raft_amd/lib/libraft_hip.so; B=raft_amd/lib/libraft_hip_head.so
python3 tools/lib_ab.py $A $B 412500 3 columns 4 2>&1 | tail -2
python3 tools/lib_ab.py $B $A 412500 3 columns 4 2>&1 | tail -2
python3 tools/lib_ab.py $A $B 206250 3 columns 4 2>&1 | tail -2
python3 tools/lib_ab.py $A $B 50000 3 columns 4 2>&1 | tail -2
one() { tag=$1; reads=$2; shift; shift; python3 bench.py --reads $reads --no-extra-legs --steps 40 --warmup 5 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 - $O/b_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("%-16s ms/step %.4f kernel %.4f pass %.4f frac %.3f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["pass_device_ms"], r["frac"]))
PY
}
for rep in 1 2 3; do one e8 412500; one e8_old 412500 --tile-bins 7552; one whole 3300000; done
python3 bench.py --reads 412500 > $O/bench_e8_full.json 2> $O/bench_e8_full.err; python3 - $O/bench_e8_full.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); e=d["e2e"]; print("e8 e2e first_pass_s %.4f after_reserve %.4f seconds %.4f" % (e["first_pass_s"], e.get("first_pass_after_reserve_s",-1), e["seconds"]))
PY
