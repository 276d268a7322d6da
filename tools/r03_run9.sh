#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_window.py tests/test_gpu_fba.py -q -k "spectral or golden or freeze or eval" 2>&1 | tail -3
bash tools/ab_bench.sh TCVOM_NO_PACK_ALL 3
