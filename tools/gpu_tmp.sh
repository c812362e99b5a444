cd $GRAFT_REPO_ROOT
line() { python - "$1" "$2" <<'PY'
import json,sys
f,tag=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(f) if l.startswith('{')][0]);r=d['roofline']
    print(tag,'ms_per_step',round(d['ms_per_step'],4),'kernel_ms',round(r['kernel_ms'],4),'pass_ms',round(r['pass_device_ms'],4),'frac',round(r['frac'],4),'pass_frac',round(r['pass_frac'],4))
except Exception as e: print(tag,'no line',e)
PY
}
for n in 50000 400000; do for v in 0 5; do
timeout 600 python bench.py --workload ultralong --reads $n --variant $v --input windows --cov-width 2 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-six-column-leg --no-packed-leg > /tmp/b.json 2> /tmp/b.err; line /tmp/b.json "ultralong $n v$v windows/2B"; tail -2 /tmp/b.err | cut -c1-300
done; done
timeout 600 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-six-column-leg --no-packed-leg > /tmp/b.json 2>/tmp/b.err; line /tmp/b.json "hg002 columns/int32"
timeout 600 python bench.py --workload ultralong --steps 5 --warmup 2 --no-e2e --no-cpu-baseline --no-six-column-leg --no-packed-leg > /tmp/b.json 2>/tmp/b.err; line /tmp/b.json "ultralong columns/int32"
