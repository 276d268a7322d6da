#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TCVOM_NT_T256X128=1 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_window.py -q -k "conv or golden" 2>&1 | tail -3
bash tools/ab_bench.sh TCVOM_NT_T256X128 3
for i in 1 2; do
a=$(python bench.py --steps 12 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
b=$(TCVOM_NT_T128=150 python bench.py --steps 12 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
c=$(TCVOM_NT_T128=150 TCVOM_NT_T256X128=1 python bench.py --steps 12 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "default $a  T128=150 $b  both $c"
done
