#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv_bn or frame_batched or row_range or residual or gemm_pair or k_major" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_window.py tests/test_gpu_syncbn.py -q -x 2>&1 | tail -3
bash tools/ab_bench.sh TCVOM_NO_BN_GROUP_CAP 3
