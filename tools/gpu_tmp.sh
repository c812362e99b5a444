cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_h
timeout 300 python tools/debug_wave.py edge_reads s300_default s200_smallparams s150_reso1 s60_ultralong s300_sym_shuffled synth 2>&1 | grep "==" | cut -c1-160
line() { python - "$1" "$2" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]);r=d['roofline']
    print(tag,'ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4),'pass_frac',round(r['pass_frac'],4))
except Exception as e: print(tag,'no line',e, open(f.replace('.json','.err')).read()[-500:])
PY
}
B="--steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-packed-leg --no-six-column-leg"
for i in 1 2; do
timeout 600 python bench.py $B --workload ultralong > gpurun_out/r04_h/bench_ul.json 2> gpurun_out/r04_h/bench_ul.err; line gpurun_out/r04_h/bench_ul.json "ultralong columns"
timeout 600 python bench.py $B > gpurun_out/r04_h/bench_full.json 2> gpurun_out/r04_h/bench_full.err; line gpurun_out/r04_h/bench_full.json "hg002 columns"
done
timeout 1500 python -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
