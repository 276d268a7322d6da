#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for d in fp16 bf16; do TCVOM_DTYPE=$d timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "scores_softmax or gemm_pair" 2>&1 | grep -E "^E|assert|passed|failed" | head -40; done
