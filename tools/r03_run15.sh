#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for d in fp16 bf16; do TCVOM_DTYPE=$d timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "k_major or scores_softmax or gca_ or gemm_pair" 2>&1 | tail -4; done
timeout 1200 python -m pytest tests/test_gpu_window.py -q -x -k "north_star or full_size or golden" 2>&1 | tail -3
bash tools/ab_bench.sh TCVOM_NO_GCA_KMAJOR_A 3
