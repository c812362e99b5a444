#!/bin/bash
# how long may the ranges of a mid-sized set be?
mkdir -p gpurun_out/r06_s20
O=gpurun_out/r06_s20
one() { tag=$1; reads=$2; shift; shift; python3 bench.py --reads $reads --no-extra-legs --steps 40 --warmup 5 "$@" > $O/b_$tag.json 2> $O/b_$tag.err; python3 - $O/b_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("%-16s ms/step %.4f kernel %.4f pass %.4f" % (sys.argv[2], d["ms_per_step"], r["kernel_ms"], r["pass_device_ms"]))
PY
}
for rep in 1 2; do
for q in 0 11904 15872 19840 23808 31744; do one e8_q$q 412500 --tile-bins $q; done
for q in 0 7936 11904 15872; do one r825k_q$q 825000 --tile-bins $q; done
for q in 0 3968 7936 11904; do one r200k_q$q 206250 --tile-bins $q; done
for q in 0 1920 3968 7936; do one r50k_q$q 50000 --tile-bins $q; done
done
for q in 0 11904 23808; do one whole_q$q 3300000 --tile-bins $q; done
