#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_switches.py -q -x -s 2>&1 | grep -E "difference|passed|failed|Error|assert" | head
