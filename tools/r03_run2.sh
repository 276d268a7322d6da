#!/bin/bash
# round 3, second measurement: fp16 default -- updated parity tests + the other-dtype subprocess, GCA chain A/B, kernel trace, ATen op map
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dtype_builds.py tests/test_gpu_window.py tests/test_gpu_index.py tests/test_gpu_syncbn.py -m gpu -q -s 2>&1 | grep -v "^$" > gpurun_out/r03_run2_tests_full.log
tail -5 gpurun_out/r03_run2_tests_full.log
bash tools/ab_bench.sh TCVOM_GCA_CHAIN 3 2>&1 | tee gpurun_out/r03_gca_chain_ab.log
cd /tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r03_a
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 10 --warmup 4 --no-cpu-baseline > $out/kt.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $db 90 > gpurun_out/r03_a_kernel_stats_1080p.md
python tools/rocpd_shapes.py $db > gpurun_out/r03_a_kernel_shapes_1080p.md 2>/dev/null
python tools/rocpd_timeline.py $db > gpurun_out/r03_a_timeline.md 2>/dev/null
tail -1 $out/kt.log | cut -c1-300
rm -f $db
timeout 600 python tools/aten_ops.py > gpurun_out/r03_aten_ops.log 2>&1
tail -60 gpurun_out/r03_aten_ops.log
