#!/bin/bash
# Same binary, same input, several processes: does the pileup kernel's time move with the process (placement) or with the box's
# clocks?  Samples rocm-smi clocks / power beside the runs.  usage: tools/variance_probe.sh [n]
cd $GRAFT_REPO_ROOT
N=${1:-6}
( while true; do echo "T $(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'fclk|mclk|sclk|Power' | sed 's/GPU\[0\]\s*: //' | tr '\n' ';')"; sleep 0.25; done ) > gpurun_out/variance_smi.txt 2>&1 &
SMI=$!
for i in $(seq 1 $N); do
  echo "RUN $i start $(date +%s.%N)"
  if [ $((i % 2)) = 1 ]; then X=--plain-input-memory; else X=; fi; echo "  $X"; python bench.py $X --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('  ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4), 'pass', round(d['roofline']['pass_device_ms'],4))"
  echo "RUN $i end $(date +%s.%N)"
done
kill $SMI
