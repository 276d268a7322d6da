#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "spectral or row_range or conv_bn_act_train or frame_batched" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_window.py -q -x 2>&1 | tail -5
bash tools/ab_bench.sh TCVOM_NO_SN_DOT 3
