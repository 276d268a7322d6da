#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r03_run7_tests.log
bash tools/profile_round.sh r03_c 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/r03_c_bench_1gpu.json
cut -c1-400 gpurun_out/r03_c_bench_1gpu.json
python bench.py --steps 10 --warmup 3 --sync-bn --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-300
