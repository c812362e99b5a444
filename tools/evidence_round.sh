#!/bin/bash
# Everything the round's DESIGN.md numbers come from, in one GPU call; summaries land in gpurun_out/<tag>/ (copy the
# ones to be judged into profiles/).  usage: tools/evidence_round.sh <tag>
TAG=${1:-r04_z}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
# 0. which box: partition modes, memory vendor, clocks (the kernel's time differs by box and by session: DESIGN.md I.4)
rocm-smi --showmemorypartition --showcomputepartition --showperflevel --showmaxpower --showmemvendor --showvbios --showclocks 2>&1 | grep -v "^=\|^$" > gpurun_out/$TAG/box.txt
# 1. the headline line (default bench), rocprofv3 kernel stats of the same command, PMC traffic (FETCH_SIZE / WRITE_SIZE in passes of their own)
tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1
# 2. SQ counters of the dominant kernel, per-dispatch timeline of a pass
tools/pmc_probe.sh 3300000 > gpurun_out/$TAG/sq_counters.txt 2>&1
tools/pass_timeline.sh ${TAG}_tl > gpurun_out/$TAG/pass_timeline.txt 2>&1
tools/pass_timeline.sh ${TAG}_tlg --input grouped > gpurun_out/$TAG/pass_timeline_grouped.txt 2>&1
tools/pass_timeline.sh ${TAG}_tlw --input windows --cov-width 1 > gpurun_out/$TAG/pass_timeline_windows_w1.txt 2>&1
# 3. what the memory system gives the pass's shape, and how much of that is where a buffer lies (hipMalloc / chunks one after the
#    other / every eighth chunk of a wide span: what the engine does); the same binary in several processes, inputs in torch's
#    memory or in the engine's; the kernel's parts switched off one at a time inside one process
[ -x tools/membench ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench > /dev/null 2>&1
tools/membench 7.4 10 32 > gpurun_out/$TAG/membench.txt 2>&1
for m in 0 2 10; do echo "== mode $m"; tools/membench 7.4 $m 32 8 2>&1 | grep buffer; done > gpurun_out/$TAG/membench_placement.txt 2>&1
tools/variance_probe.sh 6 > gpurun_out/$TAG/variance_probe.txt 2>&1
# (the run-time switches exist in a diagnostic build only: make -C raft_amd/csrc LIB=libraft_hip_diag.so BUILD=../../build/csrc_diag DEFS="-DRAFT_WAVE_DIAG")
if [ -f raft_amd/lib/libraft_hip_diag.so ]; then
  export RAFT_HIP_LIB=$PWD/raft_amd/lib/libraft_hip_diag.so
  ( echo "# six columns, int32 out; mode bits: 1 no reuse of the per-read tables, 2 no run scan, 4 no scatter, 8 no coverage stores"; python tools/mode_probe.py RAFT_WAVE_MODE=0,1,2,4,8,14 3
    echo "# six columns, a byte per window out"; PROBE_WIDTH=1 python tools/mode_probe.py RAFT_WAVE_MODE=0,1,2,4,8,14 3
    echo "# window records, a byte per window out"; PROBE_FORM=windows PROBE_WIDTH=1 python tools/mode_probe.py RAFT_WAVE_MODE=0,1,2,4,8,14 3
    echo "# workers (of 4096)"; python tools/mode_probe.py RAFT_WAVE_WAVES=4096,3072,2048,1024 2 ) 2>&1 | grep -E "RAFT_|^#" > gpurun_out/$TAG/mode_probe.txt
  unset RAFT_HIP_LIB
fi
# 4. other workloads and forms: configs[4] (ultralong), configs[1] (50 k reads), an eighth of configs[2] (one of eight GPUs in configs[3]),
#    window records in / a byte per window out as a headline of its own (with stats and traffic), general streams (shuffled, non-symmetric)
for w in ultralong s50k; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/$TAG/bench_$w.json 2> gpurun_out/$TAG/bench_$w.err; done
python bench.py --reads 412500 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/$TAG/bench_slice412k.json 2> gpurun_out/$TAG/bench_slice412k.err
python bench.py --reads 412500 $B --input windows --cov-width 1 --steps 20 --warmup 3 > gpurun_out/$TAG/bench_slice412k_windows_w1.json 2> gpurun_out/$TAG/bench_slice412k_windows_w1.err
tools/pass_timeline.sh ${TAG}_tl412 --reads 412500 > gpurun_out/$TAG/pass_timeline_slice412k.txt 2>&1
tools/pass_timeline.sh ${TAG}_tl412g --reads 412500 --input grouped > gpurun_out/$TAG/pass_timeline_slice412k_grouped.txt 2>&1
tools/pass_timeline.sh ${TAG}_tlul --workload ultralong > gpurun_out/$TAG/pass_timeline_ultralong.txt 2>&1
tools/pass_timeline.sh ${TAG}_tl50k --workload s50k > gpurun_out/$TAG/pass_timeline_s50k.txt 2>&1
tools/profile_round.sh ${TAG}_win $B --input windows --cov-width 1 > gpurun_out/$TAG/profile_round_windows.log 2>&1
tools/profile_round.sh ${TAG}_ul $B --workload ultralong > gpurun_out/$TAG/profile_round_ultralong.log 2>&1
python bench.py $B --shuffle > gpurun_out/$TAG/bench_shuffle.json 2> gpurun_out/$TAG/bench_shuffle.err
python bench.py $B --nonsym > gpurun_out/$TAG/bench_nonsym.json 2> gpurun_out/$TAG/bench_nonsym.err
tools/pass_timeline.sh ${TAG}_tlsh --shuffle > gpurun_out/$TAG/pass_timeline_shuffle.txt 2>&1
# 5. host to host: the pipeline's stage clock with the engine deriving offsets and window records itself, and with prepared input
python tools/pipe_trace.py 3300000 0 columns_d4 2> gpurun_out/$TAG/pipeline_trace_columns_d4.txt
python tools/pipe_trace.py 3300000 0 windows_d4 2> gpurun_out/$TAG/pipeline_trace_windows_d4.txt
tail -3 gpurun_out/$TAG/profile_round.log; grep -E "SQ_INSTS|SQ_WAIT_ANY|SQ_WAVE_CYCLES|BANK_CONFLICT|IDX_ACTIVE" gpurun_out/$TAG/sq_counters.txt | head -20
