#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fba.py tests/test_gpu_dim.py -q -x -k "tam" 2>&1 | tail -3
TCVOM_DTYPE=bf16 timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "tam" 2>&1 | tail -3

cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 80 | grep -E "total kernel|tam_"
grep "^{" /tmp/kt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['tam'], d['roofline']['tam_all_unknown'])"
