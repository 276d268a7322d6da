#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_full_gpu.log
tail -5 gpurun_out/r03_full_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_now.json 2> gpurun_out/r03_bench_now.err
tail -c 1500 gpurun_out/r03_bench_now.json
