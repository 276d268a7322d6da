#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "pack or spectral or wsconv or conv_fwd_bwd" 2>&1 | tail -3
TCVOM_DTYPE=bf16 timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "pack or spectral or wsconv" 2>&1 | tail -3
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 60 | grep -E "sn_|adam|total kernel"
