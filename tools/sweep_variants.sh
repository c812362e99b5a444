mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for v in ${VARIANTS:-0 1 2}; do
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --variant $v > gpurun_out/bench_v$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_v$v.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); print("variant $v", "ms/step %.3f"%j["ms_per_step"], "kernel_ms %.3f"%j["roofline"]["kernel_ms"], "frac %.3f"%j["roofline"]["frac"])
else: print("variant $v failed")
PY
done
