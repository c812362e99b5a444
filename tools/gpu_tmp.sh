cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_h
line() { python - "$1" "$2" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]);r=d['roofline']
    print(tag,'ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4),'pass_frac',round(r['pass_frac'],4), 'grouped:', {k:round(v,4) for k,v in d.get('grouped',{}).items() if k in ('ms_per_step','kernel_ms','pass_device_ms')})
except Exception as e: print(tag,'no line',e, open(f.replace('.json','.err')).read()[-500:])
PY
}
B="--steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-packed-leg"
for i in 1 2; do
timeout 600 python bench.py $B --reads 412500 > gpurun_out/r04_h/bench_slice.json 2> gpurun_out/r04_h/bench_slice.err; line gpurun_out/r04_h/bench_slice.json "slice412k columns"
timeout 600 python bench.py $B --reads 412500 --input windows --cov-width 1 --no-six-column-leg > gpurun_out/r04_h/bench_slice_w1.json 2> gpurun_out/r04_h/bench_slice_w1.err; line gpurun_out/r04_h/bench_slice_w1.json "slice412k windows/byte"
timeout 600 python bench.py $B --workload s50k > gpurun_out/r04_h/bench_s50k.json 2> gpurun_out/r04_h/bench_s50k.err; line gpurun_out/r04_h/bench_s50k.json "s50k columns"
done
timeout 600 python bench.py $B > gpurun_out/r04_h/bench_full.json 2> gpurun_out/r04_h/bench_full.err; line gpurun_out/r04_h/bench_full.json "hg002 columns"
