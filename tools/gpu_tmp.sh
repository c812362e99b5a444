cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_h
line() { python - "$1" "$2" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]);r=d['roofline']
    print(tag,'ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4),'pass_frac',round(r['pass_frac'],4), 'path', d['config']['interval_path'], d['self_check'])
except Exception as e: print(tag,'no line',e, open(f.replace('.json','.err')).read()[-500:])
PY
}
B="--steps 5 --warmup 2 --no-cpu-baseline"
timeout 600 python bench.py $B --shuffle > gpurun_out/r04_h/bench_shuffle.json 2> gpurun_out/r04_h/bench_shuffle.err; line gpurun_out/r04_h/bench_shuffle.json "hg002 shuffled"
timeout 600 python bench.py $B --nonsym > gpurun_out/r04_h/bench_nonsym.json 2> gpurun_out/r04_h/bench_nonsym.err; line gpurun_out/r04_h/bench_nonsym.json "hg002 nonsym shuffled"
timeout 600 python bench.py $B --force-bucket --no-e2e --no-packed-leg --no-six-column-leg > gpurun_out/r04_h/bench_bucket.json 2> gpurun_out/r04_h/bench_bucket.err; line gpurun_out/r04_h/bench_bucket.json "hg002 sorted, counting-sort path"
timeout 1500 python -m pytest tests/test_gpu_routed.py tests/test_gpu_consistency.py tests/test_gpu_configs.py::test_config3_full_size -x -q 2>&1 | tail -4
