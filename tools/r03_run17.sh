#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gca_attention_kernel_tight" 2>&1 | grep -E "^E|passed|failed" | head -20
