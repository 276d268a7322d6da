#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r03_run3_tests.log
bash tools/profile_round.sh r03_b 2>&1 | tail -12
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r03_b_bench_1gpu.json
cut -c1-1500 gpurun_out/r03_b_bench_1gpu.json
