#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "large_tile or conv_fwd_bwd or frame_batched or dense_gemm" 2>&1 | tail -2
bash tools/ab_bench.sh TCVOM_TT_T64=0 3
