#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_kt
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $out/kt.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $db 120 > gpurun_out/kt_latest.md
rm -f $db
grep -n "dilate\|colsum\|masked_l1\|head_conv\|preprocess\|assemble" gpurun_out/kt_latest.md | cut -c1-160
