#!/bin/bash
# round 5 closing evidence on the final sources: full GPU suite, headline + window-record + ultralong lines with stats and counters,
# general-bucketing lines, full-size bit-exact compare
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_b; mkdir -p $OUT
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -14 > $OUT/pytest_gpu.txt; tail -2 $OUT/pytest_gpu.txt
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
tools/profile_round.sh r05_b > $OUT/profile_round.log 2>&1
tools/profile_round.sh r05_b_win $B --input windows --cov-width 1 > $OUT/profile_round_windows.log 2>&1
tools/profile_round.sh r05_b_ul $B --workload ultralong > $OUT/profile_round_ultralong.log 2>&1
python3 bench.py $B --shuffle > $OUT/bench_shuffle.json 2> $OUT/bench_shuffle.err
python3 bench.py $B --nonsym > $OUT/bench_nonsym.json 2> $OUT/bench_nonsym.err
tools/pass_timeline.sh r05_b_tlsh --shuffle > $OUT/pass_timeline_shuffle.txt 2>&1
tools/pass_timeline.sh r05_b_tl > $OUT/pass_timeline.txt 2>&1
python3 bench.py --reads 412500 --no-cpu-baseline --steps 20 --warmup 3 --no-placement-ab > $OUT/bench_slice412k.json 2> $OUT/bench_slice412k.err
timeout 900 python3 tools/full_compare.py --workload hg002 > $OUT/full_compare_hg002.txt 2>&1; tail -3 $OUT/full_compare_hg002.txt
timeout 900 python3 tools/full_compare.py --workload ultralong > $OUT/full_compare_ultralong.txt 2>&1; tail -3 $OUT/full_compare_ultralong.txt
tail -4 $OUT/profile_round.log
bash tools/r05/s9.sh
