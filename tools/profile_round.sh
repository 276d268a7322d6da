#!/bin/bash
# One round's profile set on a GPU box (run through gpurun from the repo root):  tools/profile_round.sh r02
#   <tag>_kernel_stats.md / <tag>_kernel_shapes.md : rocprofv3 --kernel-trace --stats over a 10-step bench run
#   <tag>_hbm_traffic_pmc.md                       : FETCH_SIZE and WRITE_SIZE in separate --pmc passes (rocpd_pmc.py)
#   <tag>_mfma_busy.md                             : SQ_VALU_MFMA_BUSY_CYCLES pass (rocpd_mfma.py)
# Summaries land in gpurun_out/; copy the ones to keep into profiles/.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-other-configs > $out/kt.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $db 70 > gpurun_out/${tag}_kernel_stats.md
python tools/rocpd_shapes.py $db > gpurun_out/${tag}_kernel_shapes.md 2>/dev/null
( python tools/rocpd_timeline.py $db 0.5 40; python tools/rocpd_timeline.py $db step ) > gpurun_out/${tag}_timeline.md 2>&1
tail -1 $out/kt.log | cut -c1-300
rm -f $db
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-other-configs > $out/$c.log 2>&1
done
grep "^{" $out/kt.log | tail -1 > $out/bench_line.json
python tools/rocpd_pmc.py $(find $out/FETCH_SIZE -name "*.db" | head -1) $(find $out/WRITE_SIZE -name "*.db" | head -1) 40 $out/bench_line.json > gpurun_out/${tag}_hbm_traffic_pmc.md
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $out/sq -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-other-configs > $out/sq.log 2>&1
python tools/rocpd_mfma.py $(find $out/sq -name "*.db" | head -1) 30 > gpurun_out/${tag}_mfma_busy.md
find $out -name "*.db" -delete
ls -la gpurun_out/${tag}_*
