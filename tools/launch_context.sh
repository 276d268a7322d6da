cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_ctx
mkdir -p $out
rocprofv3 --kernel-trace -d $out/kt -o kt -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-profile > $out/kt.log 2>&1
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_context.py $db > gpurun_out/r06_a_launch_context.txt 2>&1
rm -f $db
tail -70 gpurun_out/r06_a_launch_context.txt
