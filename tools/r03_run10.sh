#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for d in fp16 bf16; do TCVOM_DTYPE=$d timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "scores_softmax or gca_" 2>&1 | tail -4; done
timeout 900 python -m pytest tests/test_gpu_window.py -q -k "north_star and 1088 or full_size" -s 2>&1 | grep -E "unknown-only|backward:|passed|failed"
bash tools/ab_bench.sh TCVOM_NO_FUSED_SOFTMAX 3
