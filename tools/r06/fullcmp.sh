#!/bin/bash
# full-size compares against the oracle (every read), on the final sources
mkdir -p gpurun_out/r06_fc
python3 tools/full_compare.py --workload hg002 > gpurun_out/r06_fc/full_compare_hg002.txt 2>&1; echo "hg002 rc=$?"
python3 tools/full_compare.py --workload ultralong > gpurun_out/r06_fc/full_compare_ultralong.txt 2>&1; echo "ultralong rc=$?"
RAFT_DEEP_MIN=1 python3 tools/full_compare.py --workload s50k > gpurun_out/r06_fc/full_compare_s50k_deep.txt 2>&1; echo "s50k deep rc=$?"
tail -3 gpurun_out/r06_fc/*.txt
