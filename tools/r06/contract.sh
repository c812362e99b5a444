#!/bin/bash
mkdir -p gpurun_out/r06_end
timeout 1200 python3 -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_exchange.py tests/test_gpu_speculate.py -m gpu -x -q > gpurun_out/r06_end/pytest_contract.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/r06_end/pytest_contract.txt
python3 bench.py --gpus 2 --reads 400000 --steps 5 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/r06_end/bench_gpus2_shared_b.json; echo "gpus2 rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r06_end/bench_gpus2_shared_b.json')); print(d['scaling'], round(d['ms_per_step'],1), d['config']['sharding'][-60:], d['config'].get('exchange_form'))"
