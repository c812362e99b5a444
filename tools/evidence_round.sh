#!/bin/bash
# Everything the round's DESIGN.md numbers come from, in one GPU call; summaries land in gpurun_out/<tag>/ (copy the
# ones to be judged into profiles/).  usage: tools/evidence_round.sh <tag>
TAG=${1:-r03_a}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1
tools/pmc_probe.sh -1 3300000 > gpurun_out/$TAG/sq_counters.txt 2>&1
tools/pass_timeline.sh ${TAG}_tl > gpurun_out/$TAG/pass_timeline.txt 2>&1
python tools/stamp_probe.py 3300000 > gpurun_out/$TAG/stamp_probe.txt 2>&1
[ -x tools/membench ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench > /dev/null 2>&1
tools/membench > gpurun_out/$TAG/membench.txt 2>&1
for w in ultralong s50k; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.err; done
# an eighth of the human-scale set (what one of eight GPUs holds in BASELINE configs[3]); the full-size pass on the counting-sort path
python bench.py --reads 412500 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/$TAG/bench_slice412k.json 2> gpurun_out/$TAG/bench_slice412k.err
python bench.py --input columns --handover --force-bucket --no-cpu-baseline --no-e2e --no-packed-leg > gpurun_out/$TAG/bench_counting_sort.json 2> gpurun_out/$TAG/bench_counting_sort.err
tools/pass_timeline.sh ${TAG}_tl412 --reads 412500 > gpurun_out/$TAG/pass_timeline_slice412k.txt 2>&1
tools/pass_timeline.sh ${TAG}_tlul --workload ultralong > gpurun_out/$TAG/pass_timeline_ultralong.txt 2>&1
tools/profile_round.sh ${TAG}_ul --workload ultralong > gpurun_out/$TAG/profile_round_ultralong.log 2>&1
python tools/pipe_trace.py 3300000 0 grouped 2> gpurun_out/$TAG/pipeline_trace.txt
# window records (one word per record, read ids derived in the pileup kernel): the pass as a headline of its own, stamps, counters, pipeline
python bench.py --input windows --cov-width 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg > gpurun_out/$TAG/bench_windows_w1.json 2> gpurun_out/$TAG/bench_windows_w1.err
python bench.py --input windows --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg > gpurun_out/$TAG/bench_windows_w4.json 2> gpurun_out/$TAG/bench_windows_w4.err
python tools/stamp_probe.py 3300000 windows > gpurun_out/$TAG/stamp_probe_windows.txt 2>&1
tools/profile_round.sh ${TAG}_win --input windows --cov-width 1 > gpurun_out/$TAG/profile_round_windows.log 2>&1
python tools/pipe_trace.py 3300000 0 windows 2> gpurun_out/$TAG/pipeline_trace_windows.txt
# ... and the coverage back as four-bit steps (delta4): the device pass as a headline of its own, the pipeline
python bench.py --input windows --cov-width 8 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg > gpurun_out/$TAG/bench_windows_w8.json 2> gpurun_out/$TAG/bench_windows_w8.err
python tools/pipe_trace.py 3300000 0 windows_d4 2> gpurun_out/$TAG/pipeline_trace_windows_d4.txt
tools/pass_timeline.sh ${TAG}_tl8 --input windows --cov-width 8 > gpurun_out/$TAG/pass_timeline_windows_w8.txt 2>&1
python tools/pcie_duplex.py > gpurun_out/$TAG/pcie_duplex.txt 2>&1
tail -3 gpurun_out/$TAG/profile_round.log; grep -E "SQ_INSTS|SQ_WAIT_ANY|SQ_WAVE_CYCLES|BANK_CONFLICT|IDX_ACTIVE" gpurun_out/$TAG/sq_counters.txt | head -20
