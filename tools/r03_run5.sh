#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dim.py -q 2>&1 | tail -4
TCVOM_DTYPE=bf16 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "preprocess or losses or tam or head or conv_bias or bias" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_window.py -q -k "golden or large" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-220; done
for c in fba index; do timeout 300 python bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>&1 | tail -1 | cut -c1-200; done
