#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "row_range or conv_bn_act_train or residual_gradient or frame_batched" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_window.py -q -x 2>&1 | tail -5
bash tools/ab_bench.sh TCVOM_NT_NST=3 2
bash tools/ab_bench.sh TCVOM_NT_NST=4 2
bash tools/ab_bench.sh TCVOM_NO_RANGED 3
