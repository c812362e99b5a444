#!/bin/bash
# session 2: bench contract (multi-GPU modes on one GPU), exchange tests, shuffle byte model
OUT=gpurun_out/r06_b; mkdir -p $OUT
python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_exchange.py tests/test_gpu_dist.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo pytest rc=$?; tail -15 $OUT/pytest.log
B="--no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab"
python bench.py $B --shuffle > $OUT/bench_shuffle.json 2> $OUT/bench_shuffle.err; echo shuffle rc=$?
python bench.py --gpus 2 --reads 400000 --steps 5 --warmup 2 > $OUT/bench_g2.json 2> $OUT/bench_g2.err; echo g2 rc=$?
