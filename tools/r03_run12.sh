#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "large_tile or frame_batched" 2>&1 | tail -4
bash tools/ab_bench.sh TCVOM_NT_T96=0 3
