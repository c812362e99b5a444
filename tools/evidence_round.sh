#!/bin/bash
# Everything the round's DESIGN.md numbers come from, in one GPU call; summaries land in gpurun_out/<tag>/ (copy the
# ones to be judged into profiles/).  usage: tools/evidence_round.sh <tag>
TAG=${1:-r02_m}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1
tools/pmc_probe.sh -1 3300000 > gpurun_out/$TAG/sq_counters.txt 2>&1
tools/pass_timeline.sh ${TAG}_tl > gpurun_out/$TAG/pass_timeline.txt 2>&1
python tools/stamp_probe.py 3300000 > gpurun_out/$TAG/stamp_probe.txt 2>&1
[ -x tools/membench ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench > /dev/null 2>&1
tools/membench > gpurun_out/$TAG/membench.txt 2>&1
for w in ultralong s50k; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.err; done
python tools/pipe_trace.py 2> gpurun_out/$TAG/pipeline_trace.txt
python tools/pcie_duplex.py > gpurun_out/$TAG/pcie_duplex.txt 2>&1
tail -3 gpurun_out/$TAG/profile_round.log; grep -E "SQ_INSTS|SQ_WAIT_ANY|SQ_WAVE_CYCLES|BANK_CONFLICT|IDX_ACTIVE" gpurun_out/$TAG/sq_counters.txt | head -20
