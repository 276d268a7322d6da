#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "large_tile or conv_fwd_bwd or frame_batched" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_window.py -q -x -k "golden" 2>&1 | tail -2
python bench.py --steps 12 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
