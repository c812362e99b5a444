#!/bin/bash
# closing session: the whole GPU suite, smoke(), then everything the round's numbers come from (tools/evidence_round.sh), the eighth
# of the set a few times in processes of its own, the CLI above toy size
OUT=gpurun_out/r06_end; mkdir -p $OUT
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"
bash tools/evidence_round.sh r06_end > $OUT/evidence.log 2>&1; echo "evidence rc=$?"
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
for i in 1 2 3 4; do
  python bench.py $B --reads 412500 --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('eighth ms/step',round(d['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),'pass',round(r['pass_device_ms'],4))"
  python bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']
print('whole  ms/step',round(d['ms_per_step'],4),'kernel',round(r['kernel_ms'],4),'pass',round(r['pass_device_ms'],4),'frac',round(r['frac'],3),'pass_frac',round(r['pass_frac'],3))"
done > $OUT/eighth_vs_whole.txt 2>&1
cat $OUT/eighth_vs_whole.txt
python bench.py --gpus 2 --reads 400000 --steps 5 --warmup 2 2>/dev/null | grep "^{" > $OUT/bench_gpus2_shared.json; echo "gpus2 rc=$?"
python tools/cli_big.py 500000 2>&1 | grep -v "^   PIPE" > $OUT/cli_s500k.txt; echo "cli rc=$?"
python3 tools/full_compare.py --workload hg002 > $OUT/full_compare_hg002.txt 2>&1; echo "full_compare hg002 rc=$?"; tail -1 $OUT/full_compare_hg002.txt | cut -c1-200
python3 tools/full_compare.py --workload ultralong > $OUT/full_compare_ultralong.txt 2>&1; echo "full_compare ultralong rc=$?"; tail -1 $OUT/full_compare_ultralong.txt | cut -c1-200
