#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for d in fp16 bf16; do
TCVOM_DTYPE=$d timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "softmax_backward or gca_" 2>&1 | tail -3
done
timeout 300 python tools/gemm256_bench.py 2>&1 | tail -12
for i in 1 2; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-220; done
