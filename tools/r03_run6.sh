#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_window.py -q -s -k "golden or full_size or fidelity or large" 2>&1 | grep -E "grad-norm|backward:|passed|failed|Error|assert" | cut -c1-200
TCVOM_DTYPE=bf16 timeout 600 python -m pytest tests/test_gpu_window.py -q -k "golden" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_syncbn.py -q 2>&1 | tail -2
bash tools/ab_bench.sh TCVOM_NO_TAIL_SKIP 3
