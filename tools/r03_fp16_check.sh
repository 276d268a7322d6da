#!/bin/bash
# round 3: the fp16 build -- whole GPU suite, then the headline bench in both storage types (same box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
TCVOM_DTYPE=fp16 timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r03_fp16_tests_full.log
tail -40 gpurun_out/r03_fp16_tests_full.log
grep -n "alpha MSE\|backward:\|norm ratio\|FAILED\|passed\|failed" gpurun_out/r03_fp16_tests_full.log | cut -c1-300 > gpurun_out/r03_fp16_tests.log
for d in bf16 fp16 bf16 fp16; do
TCVOM_DTYPE=$d timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-330 | tee -a gpurun_out/r03_dtype_ab.log
done
