#!/bin/bash
# stream groups in the pileup kernel (ultralong), NT stores in the derivation, steeper chunk ramp
mkdir -p gpurun_out/r06_s17
O=gpurun_out/r06_s17
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "wave or configs or windows or delta4 or packed or grouped or deep" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for i in 1 2; do
python3 bench.py --workload ultralong --no-extra-legs > $O/bench_ul_$i.json 2> $O/bench_ul_$i.err; python3 - $O/bench_ul_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("ultralong ms/step %.4f kernel %.4f frac %.3f pass %.4f product kernel %.4f" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r["pass_device_ms"], r.get("product_path_kernel_ms",0)))
PY
done
python3 bench.py > $O/bench.json 2> $O/bench.err; python3 - $O/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; e=d["e2e"]
print("hg002 ms/step %.4f kernel %.4f frac %.3f pass %.4f" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r["pass_device_ms"]))
print("e2e from_soa %.4f s (%.3e rec/s) first %.3f; prepared %.4f s" % (e["seconds"], e["records_per_s"], e["first_pass_s"], e["prepared_input"]["seconds"]))
for k in ("window_records","window_records_delta4","packed_output"):
    print(k, round(d[k]["kernel_ms"],4), round(d[k]["pass_device_ms"],4))
PY
RAFT_TRACE_PASSES=6 python3 tools/pipe_trace.py 3300000 0 columns_d4 2> $O/trace_columns_d4.txt; grep -E "^pass |held its" $O/trace_columns_d4.txt
RAFT_TRACE_PASSES=6 python3 tools/pipe_trace.py 3300000 0 windows_d4 2> $O/trace_windows_d4.txt; grep -E "^pass |held its" $O/trace_windows_d4.txt
