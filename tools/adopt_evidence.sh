#!/bin/bash
# Copies the summaries of an evidence session (tools/evidence_round.sh <tag>, merged back into gpurun_out/) into profiles/, named per
# session; the headline workload's counter traffic also as profiles/pmc_traffic.json, which bench.py reads (it checks the kernel-source
# hash and the record count, so a stale file is ignored, not trusted).  usage: tools/adopt_evidence.sh <tag>
T=${1:?tag}; G=gpurun_out/$T; P=profiles
cp $G/bench.json $P/${T}_bench.json
for w in slice412k slice412k_windows_w1 ultralong s50k shuffle nonsym; do [ -s $G/bench_$w.json ] && cp $G/bench_$w.json $P/${T}_bench_$w.json; done
for t in "" _grouped _windows_w1 _slice412k _slice412k_grouped _ultralong _s50k _shuffle; do [ -s $G/pass_timeline$t.txt ] && grep -v "at::native\|__amd_rocclr" $G/pass_timeline$t.txt | head -40 > $P/${T}_pass_timeline$t.txt; done
cp $G/pmc_traffic.json $P/${T}_pmc_traffic.json; cp $G/pmc_traffic.json $P/pmc_traffic.json
for s in _win _ul; do [ -s gpurun_out/${T}$s/pmc_traffic.json ] && cp gpurun_out/${T}$s/pmc_traffic.json $P/${T}_pmc_traffic$s.json; done
[ -s gpurun_out/${T}_win/pmc_traffic.json ] && cp gpurun_out/${T}_win/pmc_traffic.json $P/pmc_traffic_windows_w1.json
[ -s gpurun_out/${T}_ul/pmc_traffic.json ] && cp gpurun_out/${T}_ul/pmc_traffic.json $P/pmc_traffic_ultralong.json
f=$(find $G/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${T}_kernel_stats.csv
for s in _win _ul; do f=$(find gpurun_out/${T}$s/stats -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp $f $P/${T}_kernel_stats$s.csv; done
cp $G/sq_counters.txt $P/${T}_sq_counters.txt; cp $G/box.txt $P/${T}_box.txt; cp $G/variance_probe.txt $P/${T}_variance_probe.txt
cp $G/membench.txt $P/${T}_membench.txt 2>/dev/null
for m in columns_d4 windows_d4; do [ -s $G/pipeline_trace_$m.txt ] && grep -v "d2h copy" $G/pipeline_trace_$m.txt | tail -90 > $P/${T}_pipeline_trace_$m.txt; done
ls $P | grep "^${T}_" | wc -l
