#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r03_final_gpu.log
cat gpurun_out/r03_final_gpu.log
bash tools/profile_round.sh r03_g > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err
grep "^{" gpurun_out/r03_bench_final.json | tail -1 | cut -c1-200
